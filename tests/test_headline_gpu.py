"""The workloads BASELINE.json names, under test as bench.py builds them:

* configs[1] `kitti00_8kf` — bench.make_inputs("kitti00_8kf") (1224x368, 4 levels, W = 8, P = 2000, ~8.7 k dense tracker inputs) through bench.GpuJob.step,
  the very step the headline number times: 3 x (makeImages + trackNewestCoarse) + setCoarseTrackingRef + optimize(6). Window and tracked poses against the strict
  fp32 oracle on the same keyframe: < 1e-5 (BASELINE.json's bar), and against the all-fp64 oracle no farther than the fp32 oracle is (x1.5).
* configs[2] (KITTI-05, dense=1 planeOpt=1 densemap=1) — one keyframe of that configuration on the KITTI frame shape through the C-ABI only: tracker reference
  with the plane-sampled points appended on the device, tracking, optimize, planeOptimize's scale fix, SWGrayOptimize_J, setCoarseTrackingRef inputs, and the
  plane-depth map of the newest keyframe; every stage against the oracle. (No KITTI frames exist in this image: synthetic frames of that shape; the plane fits
  of the reference's PCL RANSAC are inputs.)"""
import numpy as np
import pytest

import orc
from helpers import pose_dist, tracker_inputs, true_rel_pose
from nalo_slam_amd import binding, synth

pytestmark = pytest.mark.gpu


def test_bench_headline_step_matches_oracle():
    import bench
    win, st6, trk = bench.make_inputs("kitti00_8kf")
    assert (win.w, win.h, win.W, len(win.host)) == (1224, 368, 8, 2000) and win.levels == 4
    job = bench.GpuJob(win, st6, trk, 0)
    r = bench.pose_delta_vs_oracle(job, win, st6, trk)
    assert r["tracked_max"] < 1e-5, r
    assert r["residual_decisions_differ"] <= 3 and r["residual_slots"] > 9000
    # the same keyframe on the all-fp64 oracle: the GPU is as close to it as the fp32 oracle is
    W = win.W
    _, w2c_g, _ = job.ctx.ba_get_frames()
    res = {}
    for kind in ("f32", "f64"):
        orc.lib(kind).orc_set_sum_mode(0)
        ba = orc.ba_from_window(win, kind, state6=st6)
        ba.set_options(nthreads=6, never_break=True)
        ba.optimize(6)
        res[kind] = [ba.frame(f)["worldToCam"] for f in range(W)]
    floor = max(pose_dist(a, b) for a, b in zip(res["f32"], res["f64"]))
    mine = max(pose_dist(a, b) for a, b in zip(w2c_g, res["f64"]))
    print("headline window: GPU vs fp32 oracle %.3g, GPU vs fp64 oracle %.3g, fp32 oracle vs fp64 oracle (floor) %.3g" % (r["window_max"], mine, floor))
    assert mine < max(1e-5, 1.5 * floor), (mine, floor)
    # GPU vs the strict fp32 oracle: BASELINE.json's 1e-5 wherever the fp32 floor allows it (round-2 tolerance rule, DESIGN.md 4: two fp32 evaluations of
    # this window may differ from each other by up to twice their distance to the fp64 truth)
    assert r["window_max"] < max(1e-5, 2.0 * floor), (r, floor)
    # replaying the step gives the same poses bit for bit (no float atomics anywhere on the path)
    job.step(True, keep=True)
    _, w2c_2, _ = job.ctx.ba_get_frames()
    assert np.array_equal(w2c_g, w2c_2)
    job.ctx.close()


def test_config3_dense_planeopt_densemap_keyframe():
    w, h, W, P = 1224, 368, 8, 2000
    win = synth.make_window(w=w, h=h, W=W, P=P, seed=17, n_extra=1)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    rng = np.random.RandomState(4)
    mask = np.zeros((h, w), np.float32)
    gy0 = int(0.6 * h)
    mask[gy0:h - 10, 40:w - 40] = 7.0
    bgr = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    c = binding.Context(w, h, win.K, n_slots=W + 2)
    for i in range(W + 1):
        c.frame_upload(i, win.images[i], mask=mask if i == W - 1 else None, bgr=bgr if i == W - 1 else None)
    ba = orc.ba_from_window(win, "f32", state6=st6)
    c.ba_set_window(list(range(W)), win.world_to_cam[:W], state6=st6)
    c.ba_set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
    c.ba_set_residuals(win.exists)
    # --- back-end: optimize, planeOptimize scale fix, SWGrayOptimize_J
    r_o, r_g = ba.optimize(6), c.ba_optimize(6)
    assert abs(r_g - r_o) < 1e-4 * r_o
    fr = [ba.frame(i) for i in range(W)]
    c2w_ref = synth.se3_inv(fr[W - 2]["worldToCam"])
    cam2ref = synth.se3_mul(fr[W - 2]["worldToCam"], synth.se3_inv(fr[W - 1]["worldToCam"]))
    ba.plane_scale_fix(0.98, cam2ref, c2w_ref); c.ba_plane_scale_fix(0.98, cam2ref, c2w_ref)
    cost_o, nb_o = ba.sw_gray_optimize(); cost_g, nb_g = c.ba_sw_gray_optimize()
    assert nb_g == nb_o and abs(cost_g - cost_o) < 2e-5 * cost_o
    _, w2c_g, _ = c.ba_get_frames()
    assert max(pose_dist(w2c_g[i], ba.frame(i)["worldToCam"]) for i in range(W)) < 1e-5
    # --- front-end with dense=1: reference cloud from the IN residuals of the newest keyframe + the plane-sampled points, then tracking
    st, ac, jp, en, cp = c.ba_get_residuals()
    pg = c.ba_get_points()
    m = st[:, W - 1] == 0
    Ku, Kv, nid, hdi = cp[m, W - 1, 0], cp[m, W - 1, 1], cp[m, W - 1, 2], pg["HdiF"][m]
    assert m.sum() > 1000
    c.trk_set_ref(W - 1, Ku, Kv, nid, hdi)
    trk = orc.Tracker(w, h, win.levels, win.K)
    dI_ref, _ = orc.make_images(win.images[W - 1], win.levels); dI_new, _ = orc.make_images(win.images[W], win.levels)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    plane_dir, plane_dis = np.array([0.0, -1.0, 0.02], np.float32), 1.6
    rect = [40, w - 40, gy0, h - 10]
    a_g = c.trk_append_plane_points(plane_dir, plane_dis, 7, rect)
    a_o = trk.append_plane_points(mask, plane_dir, plane_dis, 7, rect)
    assert a_g == a_o and a_g > 2000                       # the dense branch adds more points than the sparse reference holds
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, W - 1, W)) * 0.9)
    ok_g, T_g = c.trk_track(W, T0, [0, 0], [0, 0], [1, 1], c.levels - 1)[:2]
    ok_o, T_o = trk.track(dI_new, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)[:2]
    assert ok_g == ok_o and pose_dist(T_g, T_o) < 1e-5
    # --- densemap=1: the plane-depth map of the newest keyframe (ground plane in its camera frame), against the oracle's makeMap
    plane = np.array([0.0, 1.0, -0.02, -1.6], np.float32)
    c2w = synth.se3_inv(w2c_g[W - 1])
    dm = c.dense_make_map(W - 1, plane, 7.0, c2w, cap=w * h)
    O = orc.lib()
    rect_o = np.zeros(4, np.int32)
    O.orc_dense_bbox(orc.fp(mask), w, h, 7.0, orc.ip(rect_o))
    assert list(dm["rect"]) == list(rect_o)
    cap = w * h
    pu, pv = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    pid, pc, pb = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
    acc_o = np.zeros(1, np.int32)
    dI0 = orc.make_images(win.images[W - 1], 1)[0]
    n_o = O.orc_dense_make_map(orc.fp(mask), orc.fp(dI0), orc.u8p(bgr), w, h, orc.fp(plane), 7.0, orc.ip(rect_o), 1 / win.K[0], 1 / win.K[1], win.K[2], win.K[3],
                               orc.dp(np.ascontiguousarray(c2w).reshape(-1)), orc.ip(pu), orc.ip(pv), orc.fp(pid), orc.fp(pc), orc.u8p(pb), orc.ip(acc_o))
    assert dm["n"] == n_o and n_o > 30000 and dm["accept"] == acc_o[0]
    assert np.array_equal(dm["u"], pu[:n_o]) and np.array_equal(dm["v"], pv[:n_o]) and np.array_equal(dm["color"], pc[:n_o]) and np.array_equal(dm["bgr"], pb[:n_o])
    assert np.allclose(dm["idepth"], pid[:n_o], rtol=2e-6)
    c.close()
