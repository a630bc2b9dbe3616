"""No-GPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol include/nalo_gpu.h
declares; without a device the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from nalo_slam_amd import binding


def declared_symbols(header="nalo_gpu.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nalo_[a-z0-9_]+)\s*\(", txt)) - {"nalo_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = ctypes.CDLL(binding.lib_path())
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(set(binding.EXPORTS)) == [s for s in syms if s in binding.EXPORTS]
    assert set(binding.EXPORTS) == set(syms)
    io_syms = declared_symbols("nalo_io.h")
    assert len(io_syms) == 5 and not [s for s in io_syms if not hasattr(lib, s)]


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.NaloError):
        binding.Context(64, 64, (50.0, 50.0, 31.5, 31.5), n_slots=1)


def test_product_never_imports_oracle():
    """The product package and bench's GPU path must not reference oracle/ (only tests, smoke and cpu_baseline may)."""
    pkg = os.path.join(ROOT, "nalo-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import orc" not in src and "liboracle" not in src and "orc_" not in src, os.path.join(dirpath, f)
