"""Registers the hyphenated package directory `nalo-slam_amd/` under the importable name `nalo_slam_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "nalo_slam_amd" in sys.modules:
        return sys.modules["nalo_slam_amd"]
    path = os.path.join(_ROOT, "nalo-slam_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("nalo_slam_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nalo_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
