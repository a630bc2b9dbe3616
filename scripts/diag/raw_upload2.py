import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0)
c = job.ctx
for _ in range(20): job.step(True)
c.sync()
job.enable_uploads()
for _ in range(2): job.step(True, upload=True)
c.sync(); t0 = time.perf_counter()
for _ in range(50): job.step(True, upload=True)
c.sync(); print("float leg ms/step %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
job.enable_raw_uploads()
ts = []
for i in range(30):
    c.sync(); t0 = time.perf_counter(); job.step(True, upload="raw"); c.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("raw leg per-step ms:", ["%.2f" % x for x in ts])
if os.environ.get("NALO_HOST_TIMING"):
    c.close()
