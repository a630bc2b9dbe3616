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


# ------------------------------------------------------------------------------------------------ PNG / vignette / nearest resize
def _png_bytes(arr, ctype, bitdepth, filters=(0, 1, 2, 3, 4), palette=None, idat_split=3):
    """a PNG writer for the tests: arr [h, w] or [h, w, c] of uint8 / uint16 samples (sub-byte depths: values < 2^bitdepth), rows filtered with the given
    filter types in rotation (all five PNG filters get exercised), the zlib stream split over several IDAT chunks"""
    import struct, zlib
    h, w = arr.shape[:2]
    nch = 1 if arr.ndim == 2 else arr.shape[2]
    if bitdepth == 16:
        rows = [arr[y].astype(">u2").tobytes() for y in range(h)]
    elif bitdepth == 8:
        rows = [arr[y].astype(np.uint8).tobytes() for y in range(h)]
    else:
        per = 8 // bitdepth
        rows = []
        for y in range(h):
            r = arr[y].astype(np.uint8)
            pad = (-w) % per
            r = np.r_[r, np.zeros(pad, np.uint8)].reshape(-1, per)
            rows.append(bytes(int(sum(int(v) << ((per - 1 - k) * bitdepth) for k, v in enumerate(g))) for g in r))
    bpp = max(1, nch * bitdepth // 8)
    out, prev = b"", bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        f = bytearray(len(row))
        for i in range(len(row)):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) >> 1
            else:
                pp = a + b - c; pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            f[i] = (row[i] - p) & 255
        out += bytes([ft]) + bytes(f)
        prev = row
    z = zlib.compress(out, 6)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bitdepth, ctype, 0, 0, 0))
    if palette is not None:
        png += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    n = max(1, len(z) // idat_split)
    for i in range(0, len(z), n):
        png += chunk(b"IDAT", z[i:i + n])
    return png + chunk(b"IEND", b"")


def _read_png(lib, path, mode):
    w, h, ch, dep = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    data = C.c_void_p()
    lib.nalo_io_read_png.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    rc = lib.nalo_io_read_png(str(path).encode(), mode, C.byref(w), C.byref(h), C.byref(ch), C.byref(dep), C.byref(data))
    if rc:
        return rc, None
    n = w.value * h.value * ch.value
    buf = (C.c_uint16 * n).from_address(data.value) if dep.value == 16 else (C.c_uint8 * n).from_address(data.value)
    a = np.array(buf).reshape((h.value, w.value) if ch.value == 1 else (h.value, w.value, ch.value))
    lib.nalo_io_free.argtypes = [C.c_void_p]
    lib.nalo_io_free(data)
    return 0, a


def test_png_reader_all_sample_formats(lib, tmp_path):
    rng = np.random.RandomState(3)
    w, h = 37, 23                                        # odd sizes: sub-byte rows end in padding bits
    g8 = rng.randint(0, 256, (h, w)).astype(np.uint8)
    g16 = rng.randint(0, 65536, (h, w)).astype(np.uint16)
    rgb = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    rgba = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    ga = rng.randint(0, 256, (h, w, 2)).astype(np.uint8)
    pal = rng.randint(0, 256, (16, 3)).astype(np.uint8)
    idx4 = rng.randint(0, 16, (h, w)).astype(np.uint8)
    g2 = rng.randint(0, 4, (h, w)).astype(np.uint8)
    files = dict(g8=_png_bytes(g8, 0, 8), g16=_png_bytes(g16, 0, 16), rgb=_png_bytes(rgb, 2, 8), rgba=_png_bytes(rgba, 6, 8), ga=_png_bytes(ga, 4, 8),
                 pal=_png_bytes(idx4, 3, 4, palette=pal), g2=_png_bytes(g2, 0, 2))
    for k, b in files.items():
        (tmp_path / (k + ".png")).write_bytes(b)
    GRAY, BGR, UNCH = 0, 1, 2
    gray_w = lambda c: ((9798 * c[..., 0].astype(np.int64) + 19235 * c[..., 1].astype(np.int64) + 3735 * c[..., 2].astype(np.int64) + 16384) >> 15).astype(np.uint8)
    # 8-bit grey: identical in every mode that returns grey
    assert np.array_equal(_read_png(lib, tmp_path / "g8.png", GRAY)[1], g8)
    assert np.array_equal(_read_png(lib, tmp_path / "g8.png", UNCH)[1], g8)
    assert np.array_equal(_read_png(lib, tmp_path / "g8.png", BGR)[1], np.repeat(g8[..., None], 3, 2))
    # 16-bit grey: unchanged keeps the samples (vignette path), greyscale keeps the high byte
    assert np.array_equal(_read_png(lib, tmp_path / "g16.png", UNCH)[1], g16)
    assert np.array_equal(_read_png(lib, tmp_path / "g16.png", GRAY)[1], (g16 >> 8).astype(np.uint8))
    # colour: B,G,R order, integer grey weights, alpha dropped
    assert np.array_equal(_read_png(lib, tmp_path / "rgb.png", BGR)[1], rgb[..., ::-1])
    assert np.array_equal(_read_png(lib, tmp_path / "rgb.png", GRAY)[1], gray_w(rgb))
    assert np.array_equal(_read_png(lib, tmp_path / "rgba.png", BGR)[1], rgba[..., 2::-1])
    assert np.array_equal(_read_png(lib, tmp_path / "ga.png", GRAY)[1], ga[..., 0])
    # palette and sub-byte grey are expanded
    assert np.array_equal(_read_png(lib, tmp_path / "pal.png", BGR)[1], pal[idx4][..., ::-1])
    assert np.array_equal(_read_png(lib, tmp_path / "g2.png", GRAY)[1], (g2 * 85).astype(np.uint8))
    # malformed input is refused, never crashes
    (tmp_path / "bad.png").write_bytes(files["g8"][:60])
    assert _read_png(lib, tmp_path / "bad.png", GRAY)[0] == -3
    (tmp_path / "nopng.png").write_bytes(b"P5 2 2 255 abcd")
    assert _read_png(lib, tmp_path / "nopng.png", GRAY)[0] == -3
    assert _read_png(lib, tmp_path / "missing.png", GRAY)[0] == -2


def test_vignette_and_nearest_resize(lib):
    import orc
    rng = np.random.RandomState(5)
    v16 = rng.randint(20000, 65536, 5000).astype(np.uint16)
    vm, vi = np.zeros(5000, np.float32), np.zeros(5000, np.float32)
    lib.nalo_io_make_vignette.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    assert lib.nalo_io_make_vignette(v16.ctypes.data_as(C.c_void_p), 16, 5000, fpp(vm), fpp(vi)) == 0
    ref = v16.astype(np.float32) / np.float32(v16.max())
    assert np.array_equal(vm, ref) and np.array_equal(vi, np.float32(1.0) / ref) and vm.max() == 1.0
    v8 = rng.randint(60, 256, 300).astype(np.uint8)
    assert lib.nalo_io_make_vignette(v8.ctypes.data_as(C.c_void_p), 8, 300, fpp(vm), fpp(vi)) == 0
    assert np.array_equal(vm[:300], v8.astype(np.float32) / np.float32(v8.max()))
    # nearest resize: the KITTI case (1241x376 -> 1224x368) and an upscale, 1 and 3 channels, against the oracle's restatement of cv::resize
    lib.nalo_io_resize_nearest_u8.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
    for (wo, ho, w, h) in [(1241, 376, 1224, 368), (320, 240, 640, 480), (100, 37, 33, 19)]:
        for ch in (1, 3):
            src = rng.randint(0, 256, (ho, wo) if ch == 1 else (ho, wo, ch)).astype(np.uint8)
            dst = np.zeros((h, w) if ch == 1 else (h, w, ch), np.uint8)
            assert lib.nalo_io_resize_nearest_u8(src.ctypes.data_as(C.POINTER(C.c_ubyte)), wo, ho, ch, dst.ctypes.data_as(C.POINTER(C.c_ubyte)), w, h) == 0
            assert np.array_equal(dst, orc.resize_nearest_u8(src, w, h))
            # closed form: source index = floor(x * wo / w) clipped
            sx = np.minimum(np.floor(np.arange(w) * (1.0 / (w / wo))).astype(int), wo - 1); sy = np.minimum(np.floor(np.arange(h) * (1.0 / (h / ho))).astype(int), ho - 1)
            assert np.array_equal(dst, src[sy][:, sx])


def _rectify(lib, cf):
    K = np.zeros(4); rx = np.zeros((cf.h, cf.w), np.float32); ry = np.zeros((cf.h, cf.w), np.float32); pt = C.c_int(0)
    lib.nalo_io_make_rectification.argtypes = [C.POINTER(CameraFile), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    rc = lib.nalo_io_make_rectification(C.byref(cf), dp(K), fpp(rx), fpp(ry), C.byref(pt))
    return rc, K, rx, ry, pt.value


def _rectify_oracle(cf):
    import orc
    L = orc.lib()
    K = np.zeros(4); rx = np.zeros((cf.h, cf.w), np.float32); ry = np.zeros((cf.h, cf.w), np.float32); pt = np.zeros(1, np.int32)
    L.orc_make_rectification.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                         C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    pars = np.array(list(cf.pars)); oc = np.array(list(cf.out_calib), np.float32)
    rc = L.orc_make_rectification(cf.model, dp(pars), cf.w_org, cf.h_org, cf.w, cf.h, cf.rect_mode, fpp(oc), dp(K), fpp(rx), fpp(ry), pt.ctypes.data_as(C.POINTER(C.c_int)))
    return rc, K, rx, ry, int(pt[0])


def test_rectification_tables_all_models(lib, tmp_path):
    """camera.txt -> K and remapX / remapY (Undistort::readFromFile + makeOptimalK_crop + distortCoordinates, util/Undistort.cpp:637-757, 911-1292): the product's
    host code against the oracle's restatement (EQUAL: same float operations), plus properties that do not depend on either: a pinhole crop is an affine map
    whose corners land inside the original image, and a distortion model maps the rectified principal point onto the original one."""
    cases = {
        "kitti": "Pinhole 0.5812 1.9225 0.4964 0.4689 0\n1241 376\ncrop\n1224 368\n",
        "radtan": "RadTan 458.654 457.296 367.215 248.375 -0.28340811 0.07395907 0.00019359 1.76187114e-05\n752 480\ncrop\n640 480\n",
        "fov": "0.535719308086809 0.669566858850269 0.493248545285398 0.500408664348414 0.897966326944875\n1280 1024\n0.4 0.53 0.5 0.5 0\n640 480\n",
        "kb": "KannalaBrandt 380.8 380.9 320.1 239.9 -0.01 0.02 -0.03 0.004\n640 480\nnone\n640 480\n",
        "equi": "EquiDistant 190.9 190.9 254.9 256.8 0.003 0.0007 -0.002 0.0002\n512 512\ncrop\n480 480\n",
    }
    for name, txt in cases.items():
        p = tmp_path / (name + ".txt"); p.write_text(txt)
        cf = CameraFile(); assert lib.nalo_io_read_camera(str(p).encode(), C.byref(cf)) == 0
        rc, K, rx, ry, pt = _rectify(lib, cf)
        rc_o, K_o, rx_o, ry_o, pt_o = _rectify_oracle(cf)
        assert rc == 0 and rc_o == 0 and pt == pt_o == (1 if name == "kb" else 0), name
        # equal entry for entry, except where the reference's `iy < wOrg-1` slip (util/Undistort.cpp:980) keeps taps behind a landscape image: outside here
        slip = ry_o >= cf.h_org - 1
        assert np.array_equal(K, K_o) and np.array_equal(rx[~slip], rx_o[~slip]) and np.array_equal(ry[~slip], ry_o[~slip]), name
        assert (rx[slip] == -1).all() and (ry[slip] == -1).all() and (slip.any() == (name == "fov")), name
        inside = rx >= 0
        assert inside.mean() > 0.9, name                  # (the explicit FOV calibration leaves its corners outside the original image)
        assert (rx[inside] < cf.w_org - 1).all() and (ry[inside] > 0).all()
        if name == "kitti":                       # pinhole: remap = original intrinsics applied to the rectified rays: affine in (x, y)
            fx, fy, cx, cy = list(cf.pars)[:4]
            yy, xx = np.mgrid[0:cf.h, 0:cf.w]
            assert np.allclose(rx, fx * (xx - K[2]) / K[0] + cx, atol=2e-3) and np.allclose(ry, fy * (yy - K[3]) / K[1] + cy, atol=2e-3)
            assert 0 < rx[0, 0] < 20 and cf.w_org - 21 < rx[0, -1] < cf.w_org - 1          # the crop keeps (almost) the whole width
        else:                                     # the rectified principal point looks along the optical axis: it maps to the original principal point
            x0, y0 = int(round(K[2])), int(round(K[3]))
            assert abs(rx[y0, x0] - cf.pars[2]) < 1.5 * abs(cf.pars[0] / K[0]) + 1e-3 and abs(ry[y0, x0] - cf.pars[3]) < 1.5 * abs(cf.pars[1] / K[1]) + 1e-3
    # an explicit output K on a landscape sensor whose rectified view reaches below the original image: the reference's `iy < wOrg-1` slip (util/Undistort.cpp:980)
    # leaves rows hOrg-1 <= iy < wOrg-1 "valid" (it then reads behind the image); the product marks exactly those entries outside and changes nothing else
    (tmp_path / "below.txt").write_text("Pinhole 400 400 319.5 239.5 0\n640 480\n0.5 0.5 0.5 0.1 0\n640 480\n")
    cf = CameraFile(); assert lib.nalo_io_read_camera(str(tmp_path / "below.txt").encode(), C.byref(cf)) == 0
    rc, K, rx, ry, pt = _rectify(lib, cf)
    rc_o, K_o, rx_o, ry_o, pt_o = _rectify_oracle(cf)
    assert rc == 0 and rc_o == 0 and np.array_equal(K, K_o)
    slip = (ry_o >= cf.h_org - 1)                                   # what the reference keeps although the taps leave the image
    assert slip.sum() > 1000 and (ry_o[slip] < cf.w_org - 1).all()
    assert (rx[slip] == -1).all() and (ry[slip] == -1).all()
    assert np.array_equal(rx[~slip], rx_o[~slip]) and np.array_equal(ry[~slip], ry_o[~slip])
    ok = rx >= 0
    assert ok.mean() > 0.3 and (rx[ok].astype(int) + 1 < cf.w_org).all() and (ry[ok].astype(int) + 1 < cf.h_org).all()
    # the modes the reference refuses
    (tmp_path / "full.txt").write_text("EquiDistant 190.9 190.9 254.9 256.8 0.003 0.0007 -0.002 0.0002\n512 512\nfull\n512 512\n")
    cf = CameraFile(); lib.nalo_io_read_camera(str(tmp_path / "full.txt").encode(), C.byref(cf))
    assert _rectify(lib, cf)[0] == -3
    (tmp_path / "none_resized.txt").write_text("Pinhole 500 500 320 240 0\n640 480\nnone\n320 240\n")
    cf = CameraFile(); lib.nalo_io_read_camera(str(tmp_path / "none_resized.txt").encode(), C.byref(cf))
    assert _rectify(lib, cf)[0] == -3


def _camera(lib, tmp_path, name, txt):
    p = tmp_path / (name + ".txt"); p.write_text(txt)
    cf = CameraFile(); assert lib.nalo_io_read_camera(str(p).encode(), C.byref(cf)) == 0
    return cf


def test_rectification_against_closed_forms(lib, tmp_path):
    """nalo_io_make_rectification checked against values that come from NO restatement of the reference's code (the oracle's orc_undist.c follows
    the same source, so agreeing with it proves the transcription, not the geometry):
      * pinhole crop: the largest centred view of an undistorted sensor is known in closed form - every border of the table lands just inside the sensor,
      * a model with its distortion switched off IS the pinhole (RadTan with zero coefficients, FOV with omega = 0): identical K and tables,
      * the tables invert: the ANALYTIC INVERSE of each lens model (FOV: tan; equidistant / Kannala-Brandt: Newton on the theta polynomial, in float64), applied
        to a table entry, gives back the rectified pixel's viewing ray,
      * the crop is monotone: more barrel distortion squeezes more field onto the sensor, so the rectified focal length falls,
      * the crop is tight: growing the accepted view by the search's own step (0.5 %) pushes a border off the sensor."""
    # ---- pinhole: closed form
    cf = _camera(lib, tmp_path, "pin", "Pinhole 700.5 690.25 610.2 180.7 0\n1241 376\ncrop\n1224 368\n")
    rc, K, rx, ry, pt = _rectify(lib, cf)
    assert rc == 0 and pt == 0
    fx, fy, cx, cy = list(cf.pars)[:4]
    yy, xx = np.mgrid[0:cf.h, 0:cf.w]
    assert np.allclose(rx, fx * (xx - K[2]) / K[0] + cx, atol=2e-3) and np.allclose(ry, fy * (yy - K[3]) / K[1] + cy, atol=2e-3)
    # the exact largest view would put the borders ON 0 and size-1; the search stops within its 0.5 % step + the 1e-4 sampling of the centre line, strictly inside
    for lo, hi, size, f in ((rx[:, 0], rx[:, -1], cf.w_org, fx), (ry[0, :], ry[-1, :], cf.h_org, fy)):
        assert (lo > 0).all() and (hi < size - 1).all()
        assert lo.max() < 0.006 * size + 2e-4 * f + 1e-2 and hi.min() > size - 1 - 0.006 * size - 2e-4 * f - 1e-2
    # ---- a lens with its distortion switched off is the pinhole
    base = "%s 458.654 457.296 367.215 248.375 %s\n752 480\ncrop\n640 480\n"
    ref = _rectify(lib, _camera(lib, tmp_path, "pin2", base % ("Pinhole", "0")))
    rad0 = _rectify(lib, _camera(lib, tmp_path, "rad0", base % ("RadTan", "0 0 0 0")))
    assert ref[0] == 0 and rad0[0] == 0 and np.array_equal(ref[1], rad0[1]) and np.array_equal(ref[2], rad0[2]) and np.array_equal(ref[3], rad0[3])
    fov0 = _rectify(lib, _camera(lib, tmp_path, "fov0", "458.654 457.296 367.215 248.375 0\n752 480\ncrop\n640 480\n"))       # no prefix + 5 numbers = FOV, omega = 0
    assert fov0[0] == 0 and np.array_equal(ref[1], fov0[1]) and np.array_equal(ref[2], fov0[2]) and np.array_equal(ref[3], fov0[3])
    # ---- the tables invert (analytic inverse of each model, float64)
    def newton_theta(rd, k):                                    # theta (1 + k1 t^2 + k2 t^4 + k3 t^6 + k4 t^8) = rd
        t = rd.copy()
        for _ in range(30):
            t2 = t * t
            f = t * (1 + k[0] * t2 + k[1] * t2 ** 2 + k[2] * t2 ** 3 + k[3] * t2 ** 4) - rd
            df = 1 + 3 * k[0] * t2 + 5 * k[1] * t2 ** 2 + 7 * k[2] * t2 ** 3 + 9 * k[3] * t2 ** 4
            t = t - f / df
        return t
    inv_cases = {
        "fov": ("0.535719308086809 0.669566858850269 0.493248545285398 0.500408664348414 0.897966326944875\n1280 1024\ncrop\n640 480\n",
                lambda rd, k: np.tan(rd * k[0]) / (2 * np.tan(k[0] / 2))),
        "equi": ("EquiDistant 190.9 190.9 254.9 256.8 0.003 0.0007 -0.002 0.0002\n512 512\ncrop\n480 480\n", lambda rd, k: np.tan(newton_theta(rd, k))),
        "kb": ("KannalaBrandt 380.8 380.9 320.1 239.9 -0.01 0.02 -0.03 0.004\n640 480\ncrop\n600 440\n", lambda rd, k: np.tan(newton_theta(rd, k))),
    }
    for name, (txt, inverse) in inv_cases.items():
        cf = _camera(lib, tmp_path, "inv_" + name, txt)
        rc, K, rx, ry, pt = _rectify(lib, cf)
        assert rc == 0, name
        pars = np.array(list(cf.pars))
        if pars[2] < 1:                                          # relative calibration (Undistort.cpp:765-790): scaled by the sensor size, principal point - 0.5
            pars[0] *= cf.w_org; pars[1] *= cf.h_org; pars[2] = pars[2] * cf.w_org - 0.5; pars[3] = pars[3] * cf.h_org - 0.5
        ok = rx >= 0
        assert ok.mean() > 0.99, name                            # a crop: (all but rounding) every entry is on the sensor
        yy, xx = np.mgrid[0:cf.h, 0:cf.w]
        ixn, iyn = (xx - K[2]) / K[0], (yy - K[3]) / K[1]         # the rectified pixel's ray
        mx, my = (rx.astype(np.float64) - pars[2]) / pars[0], (ry.astype(np.float64) - pars[3]) / pars[1]
        rd = np.hypot(mx, my)
        sel = ok & (rd > 1e-3)
        r_back = inverse(rd[sel], pars[4:8])
        r_ray = np.hypot(ixn, iyn)[sel]
        assert np.abs(r_back - r_ray).max() < 2e-4 * max(1.0, r_ray.max()), (name, np.abs(r_back - r_ray).max())
        # and the direction is kept: the sensor offset is parallel to the ray
        cross = (mx * iyn - my * ixn)[sel] / (rd[sel] * r_ray)
        assert np.abs(cross).max() < 2e-4, name
    # ---- monotone and tight crops (barrel distortion, RadTan k1 < 0; k2 = 0.05 keeps r (1 + k1 r^2 + k2 r^4) monotone, i.e. no fold-over inside the [-5, 5) sweep:
    # a lens that folds back sends the search outside its 500 rounds, where the reference exits and the product returns NALO_IO_ERR_FORMAT - asserted last)
    fxs = []
    for k1 in (-0.1, -0.2, -0.3):
        cf = _camera(lib, tmp_path, "mono%d" % len(fxs), "RadTan 458.654 457.296 367.215 248.375 %g 0.05 0 0\n752 480\ncrop\n640 480\n" % k1)
        rc, K, rx, ry, pt = _rectify(lib, cf)
        assert rc == 0 and (rx >= 0).all()
        fxs.append((K[0], K[1]))
        # all four borders on the sensor, and the nearest border point within a few search steps of the sensor's edge on at least one axis
        slack_x = min(rx[:, 0].min(), cf.w_org - 1 - rx[:, -1].max()) / cf.w_org
        slack_y = min(ry[0, :].min(), cf.h_org - 1 - ry[-1, :].max()) / cf.h_org
        assert slack_x > 0 and slack_y > 0 and min(slack_x, slack_y) < 0.012, (k1, slack_x, slack_y)
    assert fxs[0][0] > fxs[1][0] > fxs[2][0] and fxs[0][1] > fxs[1][1] > fxs[2][1], fxs
    assert _rectify(lib, _camera(lib, tmp_path, "fold", "RadTan 458.654 457.296 367.215 248.375 -0.05 0 0 0\n752 480\ncrop\n640 480\n"))[0] == -3


def test_png_reader_survives_damaged_files(lib, tmp_path):
    """400 mutated / truncated PNGs (byte flips, cut files, overwritten header words incl. absurd sizes): every call returns a code, none crashes or throws
    across the C ABI (a damaged IHDR used to ask for a multi-gigabyte buffer)"""
    import random
    rng = np.random.RandomState(0)
    base = [_png_bytes(rng.randint(0, 256, (23, 37)).astype(np.uint8), 0, 8), _png_bytes(rng.randint(0, 65536, (23, 37)).astype(np.uint16), 0, 16),
            _png_bytes(rng.randint(0, 256, (23, 37, 3)).astype(np.uint8), 2, 8), _png_bytes(rng.randint(0, 16, (23, 37)).astype(np.uint8), 3, 4, palette=rng.randint(0, 256, (16, 3)))]
    random.seed(1)
    p = tmp_path / "f.png"
    codes = set()
    for it in range(400):
        b = bytearray(random.choice(base))
        kind = random.random()
        if kind < 0.4:
            for _ in range(random.randint(1, 6)):
                b[random.randrange(len(b))] = random.randrange(256)
        elif kind < 0.7:
            b = b[:random.randrange(8, len(b))]
        else:
            i = random.randrange(8, len(b) - 4); b[i:i + 4] = random.getrandbits(32).to_bytes(4, "big")
        p.write_bytes(bytes(b))
        rc, a = _read_png(lib, p, random.randrange(3))
        codes.add(rc)
        assert rc in (0, -3)
    # the classic: a valid file whose IHDR claims 2^31 x 2^31 pixels
    b = bytearray(base[0]); b[16:24] = (0x7fffffff).to_bytes(4, "big") * 2
    p.write_bytes(bytes(b))
    assert _read_png(lib, p, 0)[0] == -3
    assert -3 in codes
