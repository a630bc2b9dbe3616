"""Is one linearise pass bit-reproducible? (config-5 determinism test diagnosis) Runs the pass twice from a snapshot and reports which outputs differ."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from nalo_slam_amd import binding

name = sys.argv[1] if len(sys.argv) > 1 else "stress250k"
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
if P:
    bench.WORKLOADS[name]["P"] = P
win, st6, _ = bench.make_inputs(name)
W = win.W
c = binding.Context(win.w, win.h, win.K, n_slots=W)
for i in range(W):
    c.frame_upload(i, win.images[i])
c.ba_set_window(list(range(W)), win.world_to_cam[:W], state6=st6)
c.ba_set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
c.ba_set_residuals(win.exists)
c.ba_snapshot()
outs = []
for rep in range(3):
    c.ba_restore()
    e = c.ba_linearize()
    th = c.ba_get_frames()[0][W - 1].frameEnergyTH
    st, act, jp, en, cp = c.ba_get_residuals()
    HA, bA = c.ba_accumulate(0)
    acc = c.ba_get_acc13()
    outs.append(dict(e=e, th=th, st=st.copy(), act=act.copy(), jp=jp.copy(), en=en.copy(), HA=HA.copy(), acc=acc.copy()))
    print("pass %d: e=%.6f th=%r n_active=%d" % (rep, e, th, int(act.sum())))
a = outs[0]
for r, b in enumerate(outs[1:], 1):
    print("pass 0 vs %d: e equal %s, th equal %s, state equal %s (%d differ), active differ %d, energy_new differ %d, JpJdF differ %d, HA equal %s, acc13 equal %s" % (
        r, a["e"] == b["e"], a["th"] == b["th"], np.array_equal(a["st"], b["st"]), int((a["st"] != b["st"]).sum()), int((a["act"] != b["act"]).sum()),
        int((a["en"] != b["en"]).sum()), int((a["jp"] != b["jp"]).any(-1).sum()), np.array_equal(a["HA"], b["HA"]), np.array_equal(a["acc"], b["acc"])))
    d = np.argwhere(a["en"] != b["en"])
    for p, t in d[:10]:
        print("   point %d target %d host %d: energy_new %r vs %r, state %d vs %d, u,v = %.2f %.2f" % (p, t, win.host[p], a["en"][p, t], b["en"][p, t], a["st"][p, t], b["st"][p, t], win.u[p], win.v[p]))
c.close()
