"""wall time per nalo_trk_track on the headline window (A/B of tracker LM kernel variants): us per frame and per LM evaluation"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0)
job._prepare_calls()
c, L, W = job.ctx, job.ctx.L, win.W
for rep in range(3):
    n = 400
    job.evals = 0
    c.sync()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 3
        job._T[:] = job._T0[k]; job._aff[:] = 0
        c._ck(L.nalo_trk_track(c.h_, W + k, *job._trk_args, c.levels - 1, *job._trk_tail))
        job.evals += job._ne.value
    c.sync()
    dt = time.perf_counter() - t0
    print("trk_track: %.2f us per frame, %.2f evals per frame, %.3f us per evaluation" % (dt / n * 1e6, job.evals / n, dt / job.evals * 1e6))
