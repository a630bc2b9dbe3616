import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nalo_pkg; nalo_pkg.load()
import numpy as np, orc
from helpers import tracker_inputs
from nalo_slam_amd import binding, synth
win = synth.make_window(w=640, h=480, W=4, P=400, seed=7)
rng = np.random.RandomState(9)
Ku, Kv, nid, hdi = tracker_inputs(win, n=2000, seed=3)
hot_u, hot_v = rng.randint(20, win.w - 20, 60), rng.randint(20, win.h - 20, 60)
reps = rng.randint(3, 41, 60)
eu = np.concatenate([np.full(r, u) + rng.uniform(-0.45, 0.45, r) for u, r in zip(hot_u, reps)]).astype(np.float32)
ev = np.concatenate([np.full(r, v) + rng.uniform(-0.45, 0.45, r) for v, r in zip(hot_v, reps)]).astype(np.float32)
en = (10.0 ** rng.uniform(-2.5, 0.5, len(eu))).astype(np.float32)
eh = (10.0 ** rng.uniform(-7, -2, len(eu))).astype(np.float32)
perm = rng.permutation(len(Ku) + len(eu))
Ku, Kv, nid, hdi = [np.concatenate([a, b])[perm] for a, b in ((Ku, eu), (Kv, ev), (nid, en), (hdi, eh))]
c = binding.Context(win.w, win.h, win.K, n_slots=win.W + 1)
for i in range(win.W + 1): c.frame_upload(i, win.images[i])
trk = orc.Tracker(win.w, win.h, win.levels, win.K)
dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels)
trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
ia, wa = trk.get_depth(0)
runs = []
for _ in range(3):
    c.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi); runs.append(c.trk_get_depth(0))
print("run-to-run equal:", all(np.array_equal(runs[0][0], r[0]) and np.array_equal(runs[0][1], r[1]) for r in runs[1:]))
ib, wb = runs[0]
d = np.nonzero((ib != ia) | (wb != wa))[0]
print("pixels differing", len(d), "of nonzero", (wa != 0).sum())
hot = set((hot_u + win.w * hot_v).tolist())
for p in d[:15]:
    print(p, p in hot, "idepth", ib[p], ia[p], "w", wb[p], wa[p])
