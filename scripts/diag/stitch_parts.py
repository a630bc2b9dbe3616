"""ba_stitch_kernel timed per part on the headline window: top only (nalo_ba_accumulate), Schur complement only (nalo_ba_accumulate_sc). Run under
rocprofv3 --kernel-trace --stats and read the stitch rows of the trace in launch order (top, sc, top, sc, ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
win, st6, trk = bench.make_inputs(os.environ.get("WL", "kitti00_8kf"))
job = bench.GpuJob(win, st6, trk, 0)
job.step(False)
c = job.ctx
for _ in range(10):
    c.ba_linearize(False)
    c.ba_accumulate(0)
    c.ba_accumulate_sc(True)
c.sync()
c.close()
