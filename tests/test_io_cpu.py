"""On-disk formats (include/nalo_io.h): byte-exact writer output against hand-derived expectations of the reference's stream
formatting (precision 15, Eigen's aligned row vector, default-precision point lines) and parser results on the file layouts Undistort.cpp /
DatasetReader.h read. Host code only; called through the same C-ABI library."""
import ctypes as C
import os

import numpy as np
import pytest

from nalo_slam_amd import binding


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(binding.lib_path())
    return L


class CameraFile(C.Structure):
    _fields_ = [("model", C.c_int), ("n_pars", C.c_int), ("pars", C.c_double * 8), ("w_org", C.c_int), ("h_org", C.c_int), ("w", C.c_int), ("h", C.c_int),
                ("rect_mode", C.c_int), ("out_calib", C.c_float * 5)]


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def fpp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_result_txt_bytes(lib, tmp_path):
    ts = np.array([3.5, 1.25, 2.000000123456789, 4.0])              # out of order: sorted by timestamp
    valid = np.array([1, 0, 1, 0], np.uint8)                         # the earliest frame is invalid (zeros), the last repeats its predecessor
    t = np.array([[1.5, -20.25, 3.0], [9, 9, 9], [0.1, 100.0, -0.333333333333333333], [7, 7, 7]])
    q = np.array([[0, 0, 0, 1], [0.5, 0.5, 0.5, 0.5], [0.1, -0.2, 0.3, 0.927361849549570], [1, 0, 0, 0]], np.float64)
    path = str(tmp_path / "result.txt")
    assert lib.nalo_io_write_result(path.encode(), 4, dp(ts), valid.ctypes.data_as(C.POINTER(C.c_ubyte)), dp(t.ravel()), dp(q.ravel())) == 0
    lines = open(path).read().split("\n")
    assert lines[0] == "1.25 0 0 0 0 0 0 0"
    # Eigen pads the three coefficients of translation().transpose() to their common width (18 = len("-0.333333333333333"))
    assert lines[1] == "2.00000012345679                0.1                100 -0.333333333333333 0.1 -0.2 0.3 0.92736184954957"
    assert lines[2] == "3.5    1.5 -20.25      3 0 0 0 1"
    assert lines[3] == "4    1.5 -20.25      3 0 0 0 1"                  # invalid: the pose of the previous frame in the sorted history
    assert lines[4] == "" and len(lines) == 5


def test_pcd_points_bytes(lib, tmp_path):
    u, v, idp = np.array([100.0, 612.0], np.float32), np.array([50.0, 184.0], np.float32), np.array([0.5, 0.125], np.float32)
    ci = np.array([1 / 700.0, 1 / 700.0, -612 / 700.0, -184 / 700.0], np.float32)
    m = np.array([[1, 0, 0, 10], [0, 1, 0, -2], [0, 0, 1, 0.5]], np.float64)
    path = str(tmp_path / "pcl.pcd")
    for app in (0, 1):
        assert lib.nalo_io_write_pcd_points(path.encode(), app, 2, fpp(u), fpp(v), fpp(idp), fpp(ci), dp(m.ravel())) == 0
    exp = []
    for i in range(2):
        d = np.float32(1.0) / idp[i]
        x = (u[i] * ci[0] + ci[2]) * d; y = (v[i] * ci[1] + ci[3]) * d; z = d * (np.float32(1) + np.float32(2) * ci[0])       # the reference's z = depth * (1 + 2 fxi)
        w = m @ np.array([x, y, z, 1.0], np.float64)
        exp.append("%g %g %g" % tuple(w))
    assert open(path).read() == "\n".join(exp * 2) + "\n"


def test_camera_txt_forms(lib, tmp_path):
    cases = {
        "kitti": ("Pinhole 0.5812 1.9225 0.4964 0.4689 0\n1241 376\ncrop\n1224 368\n", (0, 5, -1, 1241, 376, 1224, 368)),
        "radtan": ("RadTan 458.654 457.296 367.215 248.375 -0.28340811 0.07395907 0.00019359 1.76187114e-05\n752 480\ncrop\n640 480\n", (1, 8, -1, 752, 480, 640, 480)),
        "fov_legacy": ("0.535719308086809 0.669566858850269 0.493248545285398 0.500408664348414 0.897966326944875\n1280 1024\n0.4 0.53 0.5 0.5 0\n640 480\n", (2, 5, 0, 1280, 1024, 640, 480)),
        "pinhole_legacy": ("700.5 701.5 320.25 240.75 0\n640 480\ncrop\n640 480\n", (0, 5, -1, 640, 480, 640, 480)),
        "radtan_legacy": ("458.654 457.296 367.215 248.375 -0.28 0.07 0.0002 1.7e-05\n752 480\ncrop\n640 480\n", (1, 8, -1, 752, 480, 640, 480)),
        "kb": ("KannalaBrandt 380.8 380.9 320.1 239.9 -0.01 0.02 -0.03 0.004\n640 480\nnone\n640 480\n", (4, 8, -3, 640, 480, 640, 480)),
        "equi": ("EquiDistant 190.9 190.9 254.9 256.8 0.003 0.0007 -0.002 0.0002\n512 512\nfull\n512 512\n", (3, 8, -2, 512, 512, 512, 512)),
    }
    for name, (txt, exp) in cases.items():
        p = tmp_path / (name + ".txt")
        p.write_text(txt)
        cf = CameraFile()
        assert lib.nalo_io_read_camera(str(p).encode(), C.byref(cf)) == 0, name
        assert (cf.model, cf.n_pars, cf.rect_mode, cf.w_org, cf.h_org, cf.w, cf.h) == exp, name
    # relative calibration: fx*w, fy*h, cx*w - 0.5, cy*h - 0.5 (Undistort.cpp:838-855); absolute ones are kept
    cf = CameraFile(); lib.nalo_io_read_camera(str(tmp_path / "kitti.txt").encode(), C.byref(cf))
    assert np.allclose(list(cf.pars)[:5], [0.5812 * 1241, 1.9225 * 376, 0.4964 * 1241 - 0.5, 0.4689 * 376 - 0.5, 0], rtol=0, atol=1e-12)
    cf = CameraFile(); lib.nalo_io_read_camera(str(tmp_path / "radtan.txt").encode(), C.byref(cf))
    assert list(cf.pars)[:4] == [458.654, 457.296, 367.215, 248.375] and cf.pars[7] == 1.76187114e-05
    cf = CameraFile(); lib.nalo_io_read_camera(str(tmp_path / "fov_legacy.txt").encode(), C.byref(cf))
    assert np.allclose(list(cf.out_calib), [0.4, 0.53, 0.5, 0.5, 0]) and abs(cf.pars[0] - 0.535719308086809 * 1280) < 1e-9
    (tmp_path / "bad.txt").write_text("Pinhole 1 2 3\n640 480\ncrop\n640 480\n")
    assert lib.nalo_io_read_camera(str(tmp_path / "bad.txt").encode(), C.byref(CameraFile())) == -3
    assert lib.nalo_io_read_camera(str(tmp_path / "missing.txt").encode(), C.byref(CameraFile())) == -2


def test_pcalib_and_times(lib, tmp_path):
    g = np.cumsum(np.linspace(0.5, 1.5, 256)).astype(np.float32)
    (tmp_path / "pcalib.txt").write_text(" ".join("%.9g" % x for x in g) + "\n")
    G, n = np.zeros(4096, np.float32), C.c_int(0)
    assert lib.nalo_io_read_pcalib(str(tmp_path / "pcalib.txt").encode(), 4096, fpp(G), C.byref(n)) == 0 and n.value == 256
    exp = (255.0 * (g - g[0]).astype(np.float64) / np.float64(g[-1] - g[0])).astype(np.float32)     # float difference, double scale, stored as float
    assert np.array_equal(G[:256], exp) and G[0] == 0 and G[255] == 255
    bad = g.copy(); bad[100] = bad[99]
    (tmp_path / "pcalib_bad.txt").write_text(" ".join("%.9g" % x for x in bad) + "\n")
    assert lib.nalo_io_read_pcalib(str(tmp_path / "pcalib_bad.txt").encode(), 4096, fpp(G), C.byref(n)) == -3
    (tmp_path / "pcalib_short.txt").write_text(" ".join("%d" % i for i in range(100)) + "\n")
    assert lib.nalo_io_read_pcalib(str(tmp_path / "pcalib_short.txt").encode(), 4096, fpp(G), C.byref(n)) == -3

    (tmp_path / "times.txt").write_text("00000 1.5 10.0\n00001 1.6 0\n00002 1.7 12.0\n00003 1.8\n")
    st, ex, ns, ne = np.zeros(16), np.zeros(16, np.float32), C.c_int(0), C.c_int(0)
    assert lib.nalo_io_read_times(str(tmp_path / "times.txt").encode(), 4, 16, dp(st), fpp(ex), C.byref(ns), C.byref(ne)) == 0
    assert ns.value == 4 and list(st[:4]) == [1.5, 1.6, 1.7, 1.8]
    assert ne.value == 4 and list(ex[:4]) == [10.0, 11.0, 12.0, 12.0]       # zero exposures take the mean of their positive neighbours
    assert lib.nalo_io_read_times(str(tmp_path / "times.txt").encode(), 5, 16, dp(st), fpp(ex), C.byref(ns), C.byref(ne)) == 0
    assert ns.value == 0 and ne.value == 0                                   # count mismatch: both lists dropped
    (tmp_path / "times2.txt").write_text("0 1.5\n1 1.6\n")
    assert lib.nalo_io_read_times(str(tmp_path / "times2.txt").encode(), 2, 16, dp(st), fpp(ex), C.byref(ns), C.byref(ne)) == 0
    assert ns.value == 2 and ne.value == 0                                   # no exposure anywhere: exposures dropped, stamps kept
