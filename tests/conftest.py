import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import nalo_pkg  # noqa: E402

nalo_pkg.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def small_window():
    from nalo_slam_amd import synth
    return synth.make_window(w=640, h=480, W=4, P=400, seed=7)
