"""Diagnostic (GPU box): closed-loop 1224x368 sequence, per-keyframe per-frame pose deltas and the flipped residual decisions."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nalo_pkg; nalo_pkg.load()
import numpy as np
from helpers import pose_dist
import seq_helpers as sh

w, h, n_kf = 1224, 368, 11
win, kf = sh.make_sequence(w=w, h=h, n_kf=n_kf)
B = [sh.OracleBackend(win), sh.GpuBackend(win), sh.OracleBackend(win, "f64")]
drv = sh.SequenceDriver(win, kf, B, teacher=False)

# wrap _keyframe_tail's read_points to capture residual states / energies per backend
orig_read = [b.read_points for b in B]
cap = {}
def mk(i):
    def f():
        out = orig_read[i]()
        cap[i] = out
        if i == 1:
            st, ac, jp, en, cp = B[1].c.ba_get_residuals()
            cap["gpu_en"] = en
        return out
    return f
for i, b in enumerate(B):
    b.read_points = mk(i)

recs = [drv.bootstrap()]
for k in range(2, n_kf):
    rec = drv.add_keyframe(k)
    fo, fg, f64 = rec["frames"]
    per = ["%.1e" % pose_dist(a.w2c, b.w2c) for a, b in zip(fo, fg)]
    per64 = ["%.1e" % pose_dist(a.w2c, b.w2c) for a, b in zip(fo, f64)]
    print("kf", k, "fids", rec["fids"], "flips", rec["state_mismatch"], "rmse", ["%.6f" % r for r in rec["rmse"]])
    print("   gpu-vs-f32 per frame (after marg):", per)
    print("   f64-vs-f32 per frame            :", per64)
    st0, st1, st2 = cap[0][2], cap[1][2], cap[2][2]
    d = np.argwhere(st0 != st1)
    for (p, t) in d[:10]:
        print("   flip point-row %d target col %d: oracle %d gpu %d f64 %d" % (p, t, st0[p, t], st1[p, t], st2[p, t]))
    ths = [[ "%.3f" % f.th for f in fr] for fr in rec["frames"]]
    print("   TH f32", ths[0]); print("   TH gpu", ths[1])
