"""ba_sc scope (pt_acc + SYRK) and the other BA scopes on the stress250k window and on the 1M-point window (one GPU, no hook); diagnostic for kernel variants"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
for wl in (sys.argv[1:] or ["stress250k", "shard1m"]):
    win, st6, trk = bench.make_inputs(wl)
    job = bench.GpuJob(win, st6, trk, 0)
    for _ in range(2):
        job.step(False)
    job.ctx.profile_select(None); job.ctx.profile_enable(True); job.ctx.profile_reset()
    for _ in range(3):
        job.step(False)
    job.ctx.sync()
    out = {}
    for k in ("ba_linearize", "ba_sc", "ba_reduce", "ba_resub"):
        ms, n = job.ctx.profile_get(k)
        out[k] = round(ms / max(n, 1) * 1e3, 2)
    print(wl, out, flush=True)
    job.ctx.close()
