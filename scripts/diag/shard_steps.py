"""per-keyframe wall clock of rank 0's share of an emulated 8-rank shard1m job (is the per-step time steady?)"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
gc.disable()
hook = bench.rccl_setup(None, 0, 1, "nccl")
win, st6, trk = bench.make_inputs("shard1m")
part = bench.shard(win, 0, int(os.environ.get("NALO_BENCH_EMULATE_WORLD", "8")))
job = bench.GpuJob(part, st6, trk, 0, hook)
ts = []
for i in range(16):
    job.ctx.sync(); t0 = time.perf_counter(); job.step(False); job.ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("ms per keyframe:", " ".join("%.2f" % t for t in ts))
job.ctx.close()
