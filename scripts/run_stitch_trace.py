"""Runs linearize + accumulate(top) + accumulate_sc a few times so a rocprofv3 kernel trace shows the stitch kernel per system."""
import os, sys
sys.path.insert(0, os.getcwd())
import bench
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0)
c = job.ctx
for i in range(6):
    c.ba_linearize(False)
    c.ba_accumulate(0)        # top only
    c.ba_accumulate_sc(True)  # SC only
print("ok")
