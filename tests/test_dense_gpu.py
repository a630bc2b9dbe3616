"""GPU parity of a14 (bbox scan + makeMap + the accept test with the reference's order-dependent extent quirk)."""
import numpy as np
import pytest

import orc
from nalo_slam_amd import binding

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", [(320, 240), (1224, 368), (1920, 1072)], ids=["qvga", "kitti", "stress"])     # config 3 (KITTI-05, densemap=1) and the synthetic stress frame
@pytest.mark.parametrize("case", ["ground", "far_plane_rejected", "empty", "zero_color"])
def test_make_map_matches_oracle(case, size):
    w, h = size
    rng = np.random.RandomState(3)
    img = rng.uniform(5, 250, (h, w)).astype(np.float32)
    bgr = rng.randint(0, 255, (h, w, 3)).astype(np.uint8)
    mask = np.zeros((h, w), np.float32)
    sx, sy = w / 320.0, h / 240.0
    mask[int(160 * sy):int(230 * sy), int(20 * sx):int(300 * sx)] = 7.0
    mask[int(180 * sy):int(200 * sy), int(100 * sx):int(140 * sx)] = 3.0      # a hole with another cluster value
    mask[:, 0:2] = 7.0                                # border pixels are outside the scanned range
    value = {"ground": 7.0, "far_plane_rejected": 7.0, "empty": 9.0, "zero_color": 0.0}[case]
    plane = np.array([0.0, 1.0, 0.0, -1.6], np.float32) if case != "far_plane_rejected" else np.array([0.0, 0.02, 1.0, -60.0], np.float32)
    K = (200.0 * sx, 200.0 * sx, (w - 1) / 2.0, (h - 1) / 2.0)
    c2w = np.concatenate([np.eye(3), np.array([[0.5], [0.1], [2.0]])], 1)
    ctx = binding.Context(w, h, K, n_slots=1)
    ctx.frame_upload(0, img, mask=mask, bgr=bgr)
    cap = w * h
    L = ctx.L
    rect = np.zeros(4, np.int32)
    ou, ov = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    oid, oc, ob = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
    import ctypes as C
    n, acc = C.c_int(0), C.c_int(0)
    ctx._ck(L.nalo_dense_make_map(ctx.h_, 0, binding._f(plane), value, binding._d(np.ascontiguousarray(c2w).reshape(-1)), cap, binding._i(rect),
                                  binding._i(ou), binding._i(ov), binding._f(oid), binding._f(oc), binding._u8(ob), C.byref(n), C.byref(acc)))
    # oracle
    O = orc.lib()
    dI, _ = orc.make_images(img, 1)
    rect_o = np.zeros(4, np.int32)
    O.orc_dense_bbox(orc.fp(mask), w, h, value, orc.ip(rect_o))
    assert list(rect) == list(rect_o)
    pu, pv = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    pid, pc, pb = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
    acc_o = np.zeros(1, np.int32)
    n_o = 0
    if rect_o[0] < 2 ** 30:
        n_o = O.orc_dense_make_map(orc.fp(mask), orc.fp(dI), orc.u8p(bgr), w, h, orc.fp(plane), value, orc.ip(rect_o), 1 / K[0], 1 / K[1], K[2], K[3],
                                   orc.dp(np.ascontiguousarray(c2w).reshape(-1)), orc.ip(pu), orc.ip(pv), orc.fp(pid), orc.fp(pc), orc.u8p(pb), orc.ip(acc_o))
    assert n.value == n_o
    if case == "ground":
        assert n_o > 3000 and acc.value == acc_o[0]
        if size == (320, 240):
            assert acc.value == 1
    if case == "far_plane_rejected":
        assert acc_o[0] == 0
    if n_o:
        assert acc.value == acc_o[0]
        assert np.array_equal(ou[:n_o], pu[:n_o]) and np.array_equal(ov[:n_o], pv[:n_o])       # raster order preserved
        assert np.allclose(oid[:n_o], pid[:n_o], rtol=2e-6) and np.array_equal(oc[:n_o], pc[:n_o]) and np.array_equal(ob[:n_o], pb[:n_o])
    ctx.close()


def test_make_map_box_phases_and_ragged_boxes():
    """The chunks of makeMap run over the CANDIDATES of the box ((i % 3 == 0 || j % 3 == 0), MapPoint.cpp:366) in raster order (kernels_dense.hip DnMap, round 4): every
    phase of the box origin mod 3, boxes of one or two rows / columns, and a box that fills the scanned range - point list, order, colours and the accept bit equal
    the oracle's."""
    import ctypes as C
    w, h = 320, 240
    rng = np.random.RandomState(5)
    img = rng.uniform(5, 250, (h, w)).astype(np.float32)
    bgr = rng.randint(0, 255, (h, w, 3)).astype(np.uint8)
    K = (200.0, 200.0, (w - 1) / 2.0, (h - 1) / 2.0)
    c2w = np.ascontiguousarray(np.concatenate([np.eye(3), np.array([[0.5], [0.1], [2.0]])], 1)).reshape(-1)
    plane = np.array([0.0, 1.0, 0.0, -1.6], np.float32)
    boxes = [(150 + dy, 150 + dy + 31 + dy, 40 + dx, 40 + dx + 100 + dx) for dy in range(3) for dx in range(3)]
    boxes += [(170, 172, 50, 52), (171, 173, 51, 200), (172, 230, 52, 54), (2, h - 2, 2, w - 2), (200, 203, 100, 103)]
    ctx = binding.Context(w, h, K, n_slots=1)
    O = orc.lib()
    dI, _ = orc.make_images(img, 1)
    cap = w * h
    for (y0, y1, x0, x1) in boxes:
        mask = np.zeros((h, w), np.float32)
        mask[y0:y1, x0:x1] = 7.0
        mask[(y0 + y1) // 2, x0:x1:4] = 3.0                               # some pixels of another cluster inside the box
        ctx.frame_upload(0, img, mask=mask, bgr=bgr)
        rect, rect_o = np.zeros(4, np.int32), np.zeros(4, np.int32)
        ou, ov, oid, oc, ob = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
        pu, pv, pid, pc, pb = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
        n, acc, acc_o = C.c_int(0), C.c_int(0), np.zeros(1, np.int32)
        ctx._ck(ctx.L.nalo_dense_make_map(ctx.h_, 0, binding._f(plane), 7.0, binding._d(c2w), cap, binding._i(rect), binding._i(ou), binding._i(ov), binding._f(oid),
                                          binding._f(oc), binding._u8(ob), C.byref(n), C.byref(acc)))
        O.orc_dense_bbox(orc.fp(mask), w, h, 7.0, orc.ip(rect_o))
        assert list(rect) == list(rect_o), (y0, y1, x0, x1)
        n_o = O.orc_dense_make_map(orc.fp(mask), orc.fp(dI), orc.u8p(bgr), w, h, orc.fp(plane), 7.0, orc.ip(rect_o), 1 / K[0], 1 / K[1], K[2], K[3],
                                   orc.dp(c2w), orc.ip(pu), orc.ip(pv), orc.fp(pid), orc.fp(pc), orc.u8p(pb), orc.ip(acc_o))
        assert n.value == n_o, (y0, y1, x0, x1, n.value, n_o)
        if n_o:
            assert acc.value == acc_o[0]
            assert np.array_equal(ou[:n_o], pu[:n_o]) and np.array_equal(ov[:n_o], pv[:n_o]), (y0, y1, x0, x1)
            # (the full-frame box crosses the plane's horizon row, where the plane depth cancels to ~0: an absolute floor beside the relative bound of the test above)
            assert np.allclose(oid[:n_o], pid[:n_o], rtol=2e-6, atol=2e-7) and np.array_equal(oc[:n_o], pc[:n_o]) and np.array_equal(ob[:n_o], pb[:n_o]), (y0, y1, x0, x1)
    ctx.close()
