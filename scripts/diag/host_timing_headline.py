"""headline step under NALO_HOST_TIMING=1 (set by the caller): ms per keyframe + the host-side scopes of the BA solve. usage: NALO_HOST_TIMING=1 python host_timing_headline.py"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
gc.disable()
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0)
for _ in range(20): job.step(True)
job.ctx.sync(); t0 = time.perf_counter()
for _ in range(300): job.step(True)
job.ctx.sync(); print("headline ms/KF %.4f" % ((time.perf_counter() - t0) / 300 * 1e3), flush=True)
job.ctx.close()
