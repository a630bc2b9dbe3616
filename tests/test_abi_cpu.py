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
    assert len(io_syms) == 10 and not [s for s in io_syms if not hasattr(lib, s)]


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


def test_headers_are_plain_c_and_a_c_caller_links(tmp_path):
    """the boundary is a C ABI: both headers compile as C99, and a C program linked against the library gets NALO_ERR_NO_DEVICE (not a crash, not a
    CPU fallback) from nalo_create on a machine without a GPU, and a working text writer from nalo_io.h"""
    import subprocess
    import torch
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include "nalo_gpu.h"
#include "nalo_io.h"
int main(int argc, char** argv) {
    nalo_ctx* ctx = 0;
    const float K[4] = {500.f, 500.f, 319.5f, 239.5f};
    int rc = nalo_create(&ctx, 0, 640, 480, 0, K, 2);
    printf("create=%d\n", rc);
    if (rc == 0) nalo_destroy(ctx);
    double ts[2] = {1.0, 2.0}, t[6] = {0, 0, 0, 1, 2, 3}, q[8] = {0, 0, 0, 1, 0, 0, 0, 1};
    unsigned char ok[2] = {1, 1};
    printf("write=%d\n", nalo_io_write_result(argv[1], 2, ts, ok, t, q));
    return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(binding.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lnalo_gpu", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe), str(tmp_path / "result.txt")], text=True)
    assert "write=0" in out
    assert ("create=0" in out) if torch.cuda.is_available() else ("create=%d" % binding.NALO_ERR_NO_DEVICE in out if hasattr(binding, "NALO_ERR_NO_DEVICE") else "create=-" in out)
    assert open(tmp_path / "result.txt").read() == "1 0 0 0 0 0 0 1\n2 1 2 3 0 0 0 1\n"
