"""ba_linearize on stress250k by position inside a keyframe, under three conditions: keyframes back to back; a 1.5 ms idle gap before every keyframe;
a tracked frame (latency-bound persistent kernel) before every keyframe. Diagnostic for the slow first launches of the full step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
win, st6, trk = bench.make_inputs("stress250k")
job = bench.GpuJob(win, st6, trk, 0)
for _ in range(2):
    job.step(False)
def run(mode, n=6):
    job.ctx.profile_select("ba_linearize"); job.ctx.profile_enable(True); job.ctx.profile_reset()
    for _ in range(n):
        if mode == "idle":
            job.ctx.sync(); time.sleep(0.0015)
        job.step(mode == "track")
    s = job.ctx.profile_samples("ba_linearize")
    job.ctx.profile_enable(False)
    s = s[: len(s) // 8 * 8].reshape(-1, 8)
    print("%-6s mean %.1f  by position: %s" % (mode, s.mean(), " ".join("%.1f" % v for v in s.mean(0))), flush=True)
for mode in ("b2b", "idle", "track", "b2b"):
    run(mode)
