import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0)
job.enable_uploads(); job.enable_raw_uploads()
c = job.ctx; W = win.W
def t(fn, n=200):
    fn(); c.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    c.sync(); return (time.perf_counter() - t0) / n * 1e6
print("float async upload+wait us:", t(lambda: (c.frame_upload_async(W, job._pinned[0]), c.frame_wait(W), c.sync())))
print("raw   async upload+wait us:", t(lambda: (c.frame_upload_raw_async(W, job._pinned_raw[0], exposure=1.0), c.frame_wait(W), c.sync())))
print("rebuild us:", t(lambda: (c.frame_rebuild(W), c.sync())))
for mode in (False, True, "raw"):
    job.evals = 0
    for _ in range(3): job.step(True, upload=mode)
    c.sync(); e0 = job.evals; t0 = time.perf_counter()
    for _ in range(30): job.step(True, upload=mode)
    c.sync(); dt = (time.perf_counter() - t0) / 30
    print("mode", mode, "ms/step %.3f" % (dt * 1e3), "evals/step", (job.evals - e0) / 30)
