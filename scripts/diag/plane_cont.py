"""diag: the continuation optimize() of tests/test_plane_opt_gpu.py (1224x368, W=8, P=2000) - where do GPU and oracle part?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import nalo_pkg; nalo_pkg.load()
import orc
from helpers import pose_dist
from nalo_slam_amd import synth
from test_window_state_gpu import carried_inputs, make_pair
w, h, W, P = 1224, 368, 8, 2000
win = synth.make_window(w=w, h=h, W=W, P=P, seed=21)
st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
has_prior, idz, calib_zero, aff = carried_inputs(win)
n = 8 * W + 4
ba, c = make_pair(win, st6, aff, has_prior, idz, calib_zero, np.zeros((n, n)), np.zeros(n))
r0o = ba.optimize(6); r0g = c.ba_optimize(6)
print("first optimize: rmse", r0o, r0g, "flips", int((ba.slots()[0] != c.ba_get_residuals()[0]).sum()))
fr_o = [ba.frame(i) for i in range(W)]
c2w_ref = synth.se3_inv(fr_o[W - 2]["worldToCam"])
cam2ref = synth.se3_mul(fr_o[W - 2]["worldToCam"], synth.se3_inv(fr_o[W - 1]["worldToCam"]))
ba.plane_scale_fix(1.03, cam2ref, c2w_ref); c.ba_plane_scale_fix(1.03, cam2ref, c2w_ref)
ba.sw_gray_optimize(); c.ba_sw_gray_optimize()
for its in (1, 1, 1, 1, 1, 1):
    ro = ba.optimize(1); rg = c.ba_optimize(1)
    _, w2c_g, _ = c.ba_get_frames()
    d = max(pose_dist(w2c_g[i], ba.frame(i)["worldToCam"]) for i in range(W))
    st_o, st_g = ba.slots()[0], c.ba_get_residuals()[0]
    print("optimize(1): rmse %.7f %.7f rel %.2e  pose %.2e flips %d  counts %s %s" % (ro, rg, abs(ro - rg) / ro, d, int((st_o != st_g).sum()), ba.counts(), c.ba_counts()))
c.close()
