"""Error behaviour of the C-ABI (include/nalo_gpu.h): wrong arguments and wrong call order come back as NALO_ERR_ARG / NALO_ERR_STATE with a message in
nalo_last_error - never a crash, never a kernel launched on bad operands - and the context stays usable afterwards. The reference asserts or dereferences
in these situations (e.g. CoarseTracker::trackNewestCoarse with lastRef == 0, CoarseTracker.cpp:1079); a drop-in library behind FFI must not."""
import ctypes as C

import numpy as np
import pytest

from nalo_slam_amd import binding, synth
from helpers import tracker_inputs, true_rel_pose

pytestmark = pytest.mark.gpu

ERR_ARG, ERR_STATE = -1, -4


def _last(L, h):
    return L.nalo_last_error(h).decode()


def test_create_rejects_bad_arguments():
    L = binding.load()
    h = C.c_void_p()
    K = np.array([300, 300, 160, 120], np.float32)
    assert L.nalo_create(None, 0, 320, 240, 0, binding._f(K), 2) == ERR_ARG
    assert L.nalo_create(C.byref(h), 0, 8, 240, 0, binding._f(K), 2) == ERR_ARG           # below 16 px
    assert L.nalo_create(C.byref(h), 0, 320, 240, 0, None, 2) == ERR_ARG
    assert L.nalo_create(C.byref(h), 0, 320, 240, 0, binding._f(K), 0) == ERR_ARG         # no frame slot
    assert L.nalo_create(C.byref(h), 0, 320, 240, 99, binding._f(K), 2) == ERR_ARG        # more pyramid levels than NALO_MAX_LEVELS
    assert L.nalo_create(C.byref(h), 4096, 320, 240, 0, binding._f(K), 2) == -2           # NALO_ERR_NO_DEVICE: no such device, no CPU fallback
    assert not h.value


def test_frame_and_tracker_entry_points_reject_and_recover(small_window):
    win = small_window
    c = binding.Context(win.w, win.h, win.K, n_slots=2)
    L, h = c.L, c.h_
    img = np.ascontiguousarray(win.images[win.W - 1], np.float32)
    assert L.nalo_frame_upload(h, 2, binding._f(img), None, None, None) == ERR_ARG and "nalo_frame_upload" in _last(L, h)      # slot out of range
    assert L.nalo_frame_upload(h, -1, binding._f(img), None, None, None) == ERR_ARG
    assert L.nalo_frame_upload(h, 0, None, None, None, None) == ERR_ARG
    assert L.nalo_frame_upload(None, 0, binding._f(img), None, None, None) == ERR_ARG                                          # no context
    Ku, Kv, nid, hdi = tracker_inputs(win)
    n = len(Ku)
    # the reference slot has no pyramid yet
    assert L.nalo_trk_set_ref(h, 0, n, binding._f(Ku), binding._f(Kv), binding._f(nid), binding._f(hdi)) == ERR_STATE
    assert L.nalo_trk_set_ref(h, 0, n, None, binding._f(Kv), binding._f(nid), binding._f(hdi)) == ERR_ARG
    assert L.nalo_trk_set_ref(h, 0, -5, binding._f(Ku), binding._f(Kv), binding._f(nid), binding._f(hdi)) == ERR_ARG
    # tracking without a reference
    c.frame_upload(1, win.images[win.W])
    T0 = np.ascontiguousarray(true_rel_pose(win, win.W - 1, win.W), np.float64).reshape(-1).copy()
    aff, ra, ex = np.zeros(2), np.zeros(2), np.ones(2, np.float32)
    lr, lf, mr = np.zeros(5), np.zeros(3), np.full(5, np.nan)
    ok, ne = C.c_int(0), C.c_int(0)
    rc = L.nalo_trk_track(h, 1, binding._d(T0), binding._d(aff), binding._d(ra), binding._f(ex), c.levels - 1, binding._d(mr), binding._d(lr), binding._d(lf), C.byref(ok), C.byref(ne))
    assert rc in (ERR_STATE, ERR_ARG) and _last(L, h)
    # ... and the context still works: the regular sequence runs and recovers the pose
    c.frame_upload(0, win.images[win.W - 1])
    c.trk_set_ref(0, Ku, Kv, nid, hdi)
    rc = L.nalo_trk_track(h, 7, binding._d(T0), binding._d(aff), binding._d(ra), binding._f(ex), c.levels - 1, binding._d(mr), binding._d(lr), binding._d(lf), C.byref(ok), C.byref(ne))
    assert rc in (ERR_STATE, ERR_ARG)                                                                                          # frame slot out of range
    rc = L.nalo_trk_track(h, 1, binding._d(T0), binding._d(aff), binding._d(ra), binding._f(ex), c.levels + 3, binding._d(mr), binding._d(lr), binding._d(lf), C.byref(ok), C.byref(ne))
    assert rc == ERR_ARG                                                                                                       # coarsest level beyond the pyramid
    okk, T, a2, lr2, lf2, nev = c.trk_track(1, T0.reshape(3, 4), [0, 0], [0, 0], [1, 1], c.levels - 1)
    assert okk == 1 and nev > 4
    c.close()


def test_bundle_adjustment_call_order_and_arguments(small_window):
    win = small_window
    c = binding.Context(win.w, win.h, win.K, n_slots=win.W)
    L, h = c.L, c.h_
    r = C.c_double(0)
    # nothing set
    assert L.nalo_ba_optimize(h, 6, 1, C.byref(r)) == ERR_STATE and _last(L, h)
    hostv = np.ascontiguousarray(win.host, np.int32)
    fl = [np.ascontiguousarray(x, np.float32) for x in (win.u, win.v, win.idepth, win.color, win.weights)]
    assert L.nalo_ba_set_points(h, len(hostv), binding._i(hostv), binding._f(fl[0]), binding._f(fl[1]), binding._f(fl[2]), None, binding._f(fl[3]), binding._f(fl[4]), None) == ERR_STATE
    # window size outside 2 .. NALO_MAX_WINDOW
    fs = (binding.FrameState * 32)()
    calib = np.array([win.K[0], win.K[1], win.K[2], win.K[3]], np.float64)
    assert L.nalo_ba_set_window(h, 1, fs, binding._d(calib), None) == ERR_ARG
    assert L.nalo_ba_set_window(h, 17, fs, binding._d(calib), None) == ERR_ARG
    assert L.nalo_ba_set_window(h, 4, None, binding._d(calib), None) == ERR_ARG
    # a proper window, then points with a host index outside it and a missing array
    for i in range(win.W):
        c.frame_upload(i, win.images[i])
    c.ba_set_window(list(range(win.W)), win.world_to_cam[:win.W])
    bad = hostv.copy(); bad[3] = win.W
    assert L.nalo_ba_set_points(h, len(bad), binding._i(bad), binding._f(fl[0]), binding._f(fl[1]), binding._f(fl[2]), None, binding._f(fl[3]), binding._f(fl[4]), None) == ERR_ARG
    assert "host index" in _last(L, h)
    assert L.nalo_ba_set_points(h, len(hostv), binding._i(hostv), None, binding._f(fl[1]), binding._f(fl[2]), None, binding._f(fl[3]), binding._f(fl[4]), None) == ERR_ARG
    assert L.nalo_ba_set_points(h, -1, binding._i(hostv), binding._f(fl[0]), binding._f(fl[1]), binding._f(fl[2]), None, binding._f(fl[3]), binding._f(fl[4]), None) == ERR_ARG
    # points but no residual table yet
    c.ba_set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
    assert L.nalo_ba_optimize(h, 6, 1, C.byref(r)) == ERR_STATE
    # the regular sequence still works on the same context
    c.ba_set_residuals(win.exists)
    rmse = c.ba_optimize(3, True)
    assert np.isfinite(rmse) and rmse > 0
    # marginalising a frame that is not in the window
    assert L.nalo_ba_marginalize_frame(h, win.W + 2) == ERR_ARG
    c.close()


def test_hbm_calibration_leg():
    """nalo_hbm_calibrate (the measured denominator of bench.py's roofline): plausible on an MI355X (HBM3E, 8 TB/s nominal) and rejects nonsense"""
    c = binding.Context(64, 64, (50, 50, 31.5, 31.5), n_slots=1)
    copy, triad = c.hbm_calibrate(256 << 20, 5)
    assert 1000.0 < copy < 8000.0 and 1000.0 < triad < 8000.0, (copy, triad)
    a = C.c_double(0)
    assert c.L.nalo_hbm_calibrate(c.h_, C.c_size_t(1024), 5, C.byref(a), None) == ERR_ARG          # below 1 MiB
    assert c.L.nalo_hbm_calibrate(c.h_, C.c_size_t(1 << 24), 0, C.byref(a), None) == ERR_ARG
    c.close()
