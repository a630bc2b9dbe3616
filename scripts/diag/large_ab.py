"""A/B driver: stress250k BA-only keyframe, ba_linearize mean, and the emulated N = 8 shard keyframe (same box, library variants swapped by scripts/ab.sh)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gc, torch
import bench
gc.disable()
win, st6, trk = bench.make_inputs("stress250k")
job = bench.GpuJob(win, st6, trk, 0)
for _ in range(3): job.step(False)
job.ctx.sync(); t0 = time.perf_counter()
for _ in range(12): job.step(False)
job.ctx.sync(); dt = (time.perf_counter() - t0) / 12 * 1e3
job.ctx.profile_select("ba_linearize"); job.ctx.profile_enable(True); job.ctx.profile_reset()
for _ in range(4): job.step(False)
ms, n = job.ctx.profile_get("ba_linearize")
print("stress250k BA-only ms/KF %.3f  ba_linearize %.1f us" % (dt, ms / n * 1e3), flush=True)
job.ctx.close()
os.environ["NALO_BENCH_EMULATE_WORLD"] = "8"
r = bench.shard_leg(0, 1, 0, None, torch, steps=12, warmup=3)
print("shard N=8 emulated ms/KF", r["ms_per_keyframe"], r["ba_linearize"]["avg_us"], flush=True)
