"""nalo_trk_set_ref_resident on the headline window: wall time per call (median of 300), inputs resident"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0); job._prepare_calls()
c, L, W = job.ctx, job.ctx.L, win.W
for rep in range(3):
    ts = []
    for _ in range(300):
        c.sync(); t0 = time.perf_counter(); c._ck(L.nalo_trk_set_ref_resident(c.h_, W - 1)); c.sync(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("set_ref_resident + drain: median %.1f us, p10 %.1f" % (np.median(ts), np.percentile(ts, 10)), flush=True)
