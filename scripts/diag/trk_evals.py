"""LM evaluations per tracked frame of the headline keyframe: device (nalo_trk_track) vs the strict fp32 and the all-fp64 oracle (orc_trk_counter), and pose distances"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench, orc
from nalo_slam_amd import synth
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0)
job._prepare_calls()
c, L, W = job.ctx, job.ctx.L, win.W
res = {}
for kind in ("f32", "f64"):
    t = orc.Tracker(win.w, win.h, win.levels, win.K, kind)
    t.set_ref(orc.make_images(win.images[W - 1], win.levels, kind)[0], *trk)
    out = []
    for k in range(3):
        n0 = t.L.orc_trk_counter(t.h_, 0)
        ok, T = t.track(orc.make_images(win.images[W + k], win.levels, kind)[0], job.T_init[k], [0, 0], [0, 0], [1, 1], win.levels - 1)[:2]
        out.append((t.L.orc_trk_counter(t.h_, 0) - n0, T))
    res[kind] = out
for k in range(3):
    job._T[:] = job._T0[k]; job._aff[:] = 0
    c._ck(L.nalo_trk_track(c.h_, W + k, *job._trk_args, c.levels - 1, *job._trk_tail))
    Tg = job._T.reshape(3, 4).copy()
    d = lambda A, B: float(np.linalg.norm(orc.se3_log(synth.se3_mul(A, synth.se3_inv(B)))))
    print("frame %d: evals gpu %d / f32 %d / f64 %d; |gpu - f32| %.2e, |f64 - f32| %.2e" % (k, job._ne.value, res["f32"][k][0], res["f64"][k][0], d(Tg, res["f32"][k][1]), d(res["f64"][k][1], res["f32"][k][1])))
