"""BA-only timing of the stress250k window (the stress leg of bench.py) with the host-side accounting switched on (NALO_HOST_TIMING=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
win, st6, trk = bench.make_inputs("stress250k")
job = bench.GpuJob(win, st6, trk, 0)
for _ in range(3):
    job.step(False)
job.ctx.sync()
n = int(os.environ.get("N", "10"))
t0 = time.perf_counter()
for _ in range(n):
    job.step(False)
job.ctx.sync()
dt = (time.perf_counter() - t0) / n
print("BA only: %.3f ms per keyframe = %.1f KF/s" % (dt * 1e3, 1 / dt))
job.ctx.close()
