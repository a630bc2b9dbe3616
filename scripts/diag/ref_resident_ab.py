"""same process, alternating: the headline step with the tracking reference's inputs resident on the device (nalo_trk_set_ref_resident) vs handed over as host arrays
per keyframe (nalo_trk_set_ref): ms per keyframe"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
gc.disable()
win, st6, trk = bench.make_inputs("kitti00_8kf")
job = bench.GpuJob(win, st6, trk, 0)
for _ in range(20): job.step(True)
c, L, W = job.ctx, job.ctx.L, win.W
class HostArrays:                      # stands in for the library handle: the resident call becomes the host-array call
    def __init__(self, L): self.__dict__["L"] = L
    def __getattr__(self, k):
        if k == "nalo_trk_set_ref_resident":
            return lambda h, slot: self.L.nalo_trk_set_ref(h, slot, len(job._ref[0]), *job._ref_args)
        return getattr(self.L, k)
real = c.L
for rep in range(3):
    for name, lib in (("resident", real), ("host arrays", HostArrays(real))):
        c.L = lib
        job.ctx.sync(); t0 = time.perf_counter()
        for _ in range(300): job.step(True)
        job.ctx.sync(); print("%-12s ms/KF %.4f" % (name, (time.perf_counter() - t0) / 300 * 1e3), flush=True)
c.L = real
