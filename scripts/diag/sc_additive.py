"""is H_sc additive over shards without any hook? (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from nalo_slam_amd import binding, synth
from test_shard_gpu import make_ctx
win = synth.make_window(w=640, h=480, W=4, P=900, seed=21)
st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
def run(w):
    c = make_ctx(w, st6)
    e = c.ba_linearize(); HA, bA = c.ba_accumulate(0); Hs, bs = c.ba_accumulate_sc(True)
    c.close()
    return e, HA, Hs
e, HA, Hs = run(win)
parts = [run(bench.shard(win, r, 2)) for r in range(2)]
print("E", e, parts[0][0] + parts[1][0])
print("HA rel", np.abs(parts[0][1] + parts[1][1] - HA).max() / np.abs(HA).max())
print("Hs rel", np.abs(parts[0][2] + parts[1][2] - Hs).max() / np.abs(Hs).max())
# the newest frame's threshold differs between a shard alone and the window: use explicit same TH by disabling? report diag
print("Hs diag full", np.diag(Hs)[:8]); print("Hs diag sum ", np.diag(parts[0][2] + parts[1][2])[:8])

# hooked: two threads, capture each rank's block before the sum
import threading, torch
from test_shard_gpu import _Ptr
world = 2
bar = threading.Barrier(world)
bufs, cap, out = [None] * world, [dict() for _ in range(world)], [None] * world
def rank_job(r):
    c = make_ctx(bench.shard(win, r, world), st6)
    def hook(ptr, n):
        t = torch.as_tensor(_Ptr(ptr, n), device="cuda")
        bufs[r] = t.cpu()
        cap[r].setdefault(n, []).append(bufs[r].numpy().copy())
        bar.wait(); tot = bufs[0] + bufs[1]; bar.wait()
        t.copy_(tot); torch.cuda.synchronize()
    c.ba_set_allreduce(hook)
    e = c.ba_linearize(); HA, bA = c.ba_accumulate(0); Hs, bs = c.ba_accumulate_sc(True)
    out[r] = (e, HA, Hs)
    c.ba_set_allreduce(None); c.close()
ts = [threading.Thread(target=rank_job, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
n1 = 8 * win.W + 5
blk = n1 * n1
for r in range(2):
    pre = cap[r][blk][0].reshape(n1, n1)[:n1 - 1, :n1 - 1]
    print("rank", r, "pre-sum SC block vs unhooked Hs_r:", np.abs(pre - parts[r][2]).max() / np.abs(parts[r][2]).max(), "calls", {k: len(v) for k, v in cap[r].items()})
print("hooked Hs vs full", np.abs(out[0][2] - Hs).max() / np.abs(Hs).max(), " HA", np.abs(out[0][1] - HA).max() / np.abs(HA).max())
