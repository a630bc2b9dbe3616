"""Diagnostic (GPU box): one calcResAndGS + doStep of the initialiser, device vs oracle, actual error levels."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nalo_pkg; nalo_pkg.load()
import numpy as np, orc
from helpers import rel_err, pose_dist
from nalo_slam_amd import binding, synth

w, h = 640, 480
win = synth.make_window(w=w, h=h, W=2, P=20, seed=3, n_extra=3, step_z=0.15, yaw_deg=0.1)
rp, _ = orc.pixsel_libc_tables(w * h)
c = binding.Context(w, h, win.K, n_slots=4)
for i in range(4):
    c.frame_upload(i, win.images[i])
c.pixsel_set_random(rp)
num, sf = c.init_set_first(0)
for kind in ("f32", "f64"):
    dI0, _ = orc.make_images(win.images[0], win.levels, kind); dI1, _ = orc.make_images(win.images[1], win.levels, kind)
    L = orc.lib(kind)
    offs = [L.orc_pyr_offset(w, h, l) for l in range(win.levels + 1)]
    for lvl in (win.levels - 1, 0):
        P = c.init_points(lvl)
        n = len(P["u"])
        rng = np.random.RandomState(1)
        pts = dict(u=P["u"], v=P["v"], idepth_new=(1 + 0.05 * rng.randn(n)).astype(np.float32), iR=np.ones(n, np.float32), isGood=np.ones(n, np.uint8),
                   energy=np.zeros((n, 2), np.float32), outlierTH=P["outlierTH"], lastHessian_new=np.zeros(n, np.float32), Jb=np.zeros((n, 10), np.float32))
        T = orc.se3_exp(np.array([0.002, -0.001, -0.03, 0.0005, -0.0007, 0.0002]))
        wl, hl = w >> lvl, h >> lvl
        K4 = [win.K[0] / 2 ** lvl, win.K[1] / 2 ** lvl, (win.K[2] + 0.5) / 2 ** lvl - 0.5, (win.K[3] + 0.5) / 2 ** lvl - 0.5]
        o = orc.init_calc_res_and_gs(dI0[offs[lvl]:offs[lvl + 1]], dI1[offs[lvl]:offs[lvl + 1]], wl, hl, K4, T, [0.0, 0.0], pts, kind=kind)
        g = c.init_calc_res_and_gs(0, 1, lvl, T, [0.0, 0.0], pts)
        print(kind, "lvl", lvl, "n", n, "H", rel_err(g["H"], o["H"]), "b", rel_err(g["b"], o["b"]), "Hsc", rel_err(g["Hsc"], o["Hsc"]), "bsc", rel_err(g["bsc"], o["bsc"]),
              "E3", g["E3"], o["E3"], "good mismatch", int((g["isGood_new"] != o["isGood_new"]).sum()), "Jb", rel_err(g["Jb"], o["Jb"]), "en", rel_err(g["energy_new"], o["energy_new"]))
        Hl_g = g["H"] - g["Hsc"]; Hl_o = o["H"] - o["Hsc"]
        print("     H-Hsc", rel_err(Hl_g, Hl_o), "cancellation |H|/|H-Hsc|", np.abs(o["H"]).max() / np.abs(Hl_o).max())

# per-point error distributions at level 0: GPU vs fp32 oracle, GPU vs fp64 oracle, fp32 vs fp64 oracle
lvl = 0
P = c.init_points(lvl); n = len(P["u"]); rng = np.random.RandomState(1)
pts = dict(u=P["u"], v=P["v"], idepth_new=(1 + 0.05 * rng.randn(n)).astype(np.float32), iR=np.ones(n, np.float32), isGood=np.ones(n, np.uint8),
           energy=np.zeros((n, 2), np.float32), outlierTH=P["outlierTH"], lastHessian_new=np.zeros(n, np.float32), Jb=np.zeros((n, 10), np.float32))
T = orc.se3_exp(np.array([0.002, -0.001, -0.03, 0.0005, -0.0007, 0.0002]))
K4 = list(win.K)
res = {}
for kind in ("f32", "f64"):
    dI0, _ = orc.make_images(win.images[0], win.levels, kind); dI1, _ = orc.make_images(win.images[1], win.levels, kind)
    res[kind] = orc.init_calc_res_and_gs(dI0[:w * h], dI1[:w * h], w, h, K4, T, [0.0, 0.0], pts, kind=kind)
res["gpu"] = c.init_calc_res_and_gs(0, 1, lvl, T, [0.0, 0.0], pts)
def q(a, b, name):
    a = np.asarray(a, np.float64).reshape(n, -1); b = np.asarray(b, np.float64).reshape(n, -1)
    d = np.abs(a - b).max(1) / np.maximum(np.abs(b).max(1), 1e-30)
    print("   %-12s q50 %.2e q90 %.2e q99 %.2e max %.2e  signed mean %.2e" % (name, *np.quantile(d, [0.5, 0.9, 0.99, 1.0]), ((a - b).sum(1) / np.maximum(np.abs(b).sum(1), 1e-30)).mean()))
for x, y in (("gpu", "f32"), ("gpu", "f64"), ("f32", "f64")):
    print(x, "vs", y)
    for k in ("energy_new", "Jb", "maxstep", "lastHessian_new"):
        q(res[x][k], res[y][k], k)
    for k in ("H", "b", "Hsc", "bsc"):
        print("   %-4s rel %.2e" % (k, rel_err(res[x][k], res[y][k])))
# frame download check: level-0 texels identical?
g0 = c.frame_download(1, 0)[0]
dI1, _ = orc.make_images(win.images[1], win.levels, "f32")
print("level-0 new-frame texels equal:", np.array_equal(np.asarray(g0).reshape(-1, 3), dI1[:w * h]))
