"""Regenerates tests/golden/*.npz from the CPU oracle (fp32-pointwise build, fp64 sums) on tiny seeded windows.
The reference ships no golden vectors for this path (parity unpinned); these fixtures pin the ORACLE's behaviour so that
any later change to oracle or kernels is caught, and let the GPU tests run against committed data only.
Run:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nalo_pkg  # noqa: E402

nalo_pkg.load()
import orc  # noqa: E402
from helpers import tracker_inputs, true_rel_pose  # noqa: E402
from nalo_slam_amd import synth  # noqa: E402

GOLDEN_WINDOW = dict(w=320, h=240, W=3, P=64, seed=11)
GOLDEN_IMM_WINDOW = dict(w=320, h=240, W=3, P=64, seed=12, n_extra=2, step_z=0.25, yaw_deg=0.4)


def main():
    orc.lib().orc_set_sum_mode(0)
    win = synth.make_window(**GOLDEN_WINDOW)
    out = {}
    # a1: pyramid checksums + a few texels
    dI, ab = orc.make_images(win.images[1], win.levels)
    out["pyr_sum"] = np.array([dI[:, k].astype(np.float64).sum() for k in range(3)] + [ab.astype(np.float64).sum()])
    out["pyr_texels"] = dI[[1000, 5000, 20000, 76800 + 100, 76800 + 19200 + 50]]
    # a2-a4: tracker
    Ku, Kv, nid, hdi = tracker_inputs(win, n=800, seed=2)
    trk = orc.Tracker(win.w, win.h, win.levels, win.K)
    dref = orc.make_images(win.images[win.W - 1], win.levels)[0]
    dnew = orc.make_images(win.images[win.W], win.levels)[0]
    trk.set_ref(dref, Ku, Kv, nid, hdi)
    out["trk_in"] = np.stack([Ku, Kv, nid, hdi])
    out["pc_n"] = np.array([len(trk.get_pc(l)[0]) for l in range(win.levels)])
    out["pc0_head"] = np.stack([a[:16] for a in trk.get_pc(0)])
    T = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.9)
    aff = np.array([0.97, 2.0], np.float32)
    out["trk_T"], out["trk_aff"] = T, aff
    for lvl in (0, 2):
        out["trk_stats%d" % lvl] = trk.calc_res(dnew, lvl, T, aff, 20.0)
        H, b = trk.calc_gs(lvl, float(aff[0]), 0.5)
        out["trk_H%d" % lvl], out["trk_b%d" % lvl] = H, b
    ok, Tt, afft, lr, lf = trk.track(dnew, T, [0, 0], [0, 0], [1, 1], win.levels - 1)
    out["trk_track_T"], out["trk_track_aff"], out["trk_track_ok"] = Tt, afft, np.array([ok])
    # a5-a13: BA
    st6 = synth.perturbed_poses(win, sigma_t=0.003, sigma_r=0.0003)
    out["ba_state6"] = st6
    ba = orc.ba_from_window(win, "f32", state6=st6)
    out["ba_energy"] = np.array([ba.linearize_all(False)])
    ba.apply_res()
    st, ac, jp, en = ba.slots()
    out["ba_state"], out["ba_active"], out["ba_JpJdF"] = st, ac, jp
    HA, bA, h13 = ba.accumulate(0, True)
    Hs, bs = ba.accumulate_sc(True)
    out["ba_HA"], out["ba_bA"], out["ba_Hsc"], out["ba_bsc"], out["ba_acc13"] = HA, bA, Hs, bs, h13
    out["ba_J16"] = np.stack([ba.residual(p, t)["J"] for p, t in zip(*np.nonzero(ac)) if True][:16])
    out["ba_x0"] = ba.solve_system(0)
    out["ba_step"] = ba.points()["step"]
    ba2 = orc.ba_from_window(win, "f32", state6=st6)
    out["ba_opt_rmse"] = np.array([ba2.optimize(6)])
    out["ba_opt_w2c"] = np.stack([ba2.frame(f)["worldToCam"] for f in range(win.W)])
    out["ba_opt_idepth"] = ba2.points()["idepth"]
    out["ba_opt_calib"] = ba2.calib()
    np.savez_compressed(os.path.join(HERE, "golden_r01.npz"), **out)
    print("wrote golden_r01.npz with", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "golden_r01.npz")), "bytes")


def imm_inputs(win):
    from imm_helpers import imm_points, host_to_new
    u, v, host = imm_points(win, per_host=120, seed=5)
    return u, v, host, host_to_new


def main_imm():
    """SURVEY 8(f) rank 1 (immature points): constructor outputs, two tracing rounds and the activation of a small seeded point set."""
    win = synth.make_window(**GOLDEN_IMM_WINDOW)
    W = win.W
    u, v, host, host_to_new = imm_inputs(win)
    n = len(u)
    dI = [orc.make_images(win.images[i], 1)[0] for i in range(W + 2)]
    out = {"u": u, "v": v, "host": host}
    color, weights, gradH, eth = [np.zeros((n, k), np.float32) for k in (8, 8, 3)] + [np.zeros(n, np.float32)]
    for h in range(W):
        m = host == h
        color[m], weights[m], gradH[m], eth[m] = orc.imm_create(dI[h], win.w, win.h, u[m], v[m])
    out.update(color=color, weights=weights, gradH=gradH, energyTH=eth)
    st = [np.zeros(n, np.float32), np.full(n, np.nan, np.float32), np.full(n, 5, np.int32), np.full(n, 10000, np.float32)]
    uf, vf = u.astype(np.float32), v.astype(np.float32)
    for r, new in enumerate((W, W + 1)):
        KRKi, Kt, aff = host_to_new(win, new)
        res = orc.imm_trace(dI[new], win.w, win.h, uf, vf, color, weights, gradH, eth, host, KRKi, Kt, aff, *st)
        for name, a in zip(("idmin", "idmax", "status", "quality", "lastUV", "lastInterval"), res):
            out["trace%d_%s" % (r, name)] = a
        st = list(res[:4])
    ba = orc.ba_from_window(win, "f32")
    Rt, af = ba.precalc_rt()
    idmin = np.where(np.isfinite(st[1]) & (st[2] == 0), st[0], np.float32(0.05)).astype(np.float32)
    idmax = np.where(np.isfinite(st[1]) & (st[2] == 0), st[1], np.float32(0.15)).astype(np.float32)
    out["opt_idmin"], out["opt_idmax"] = idmin, idmax
    out["opt_result"], out["opt_idepth"], out["opt_res_in"] = orc.imm_optimize(dI[:W], win.w, win.h, win.K, Rt, af, host, uf, vf, color, weights, eth, idmin, idmax, 1)
    np.savez_compressed(os.path.join(HERE, "golden_imm_r01.npz"), **out)
    print("wrote golden_imm_r01.npz with", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "golden_imm_r01.npz")), "bytes")


GOLDEN_PIXSEL = dict(w=320, h=224, W=2, P=20, seed=17, n_extra=0)


def pixsel_inputs():
    """frame (with a stripe patch: dy == 0 exactly), lidar-style mask, and the two libc streams the reference draws (stored: the fixture must not depend
    on the libc of the machine that replays it)"""
    win = synth.make_window(**GOLDEN_PIXSEL)
    img = win.images[1].copy()
    xs = np.arange(100, 220)
    img[60:130, xs] = (100 + 60 * ((xs // 3) % 2)).astype(np.float32)[None, :]
    h, w = img.shape
    mask = np.zeros((h, w), np.float32)
    mask[h // 3:, w // 5:] = np.random.RandomState(5).randint(0, 200, size=(h - h // 3, w - w // 5)).astype(np.float32)
    return win, img, mask


def main_pixsel():
    """SURVEY 8(f) rank 3 (pixel selector): block thresholds, select at two potentials, makeMaps with adaptation + sub-selection, makeMaps_lidar."""
    win, img, mask = pixsel_inputs()
    h, w = img.shape
    rp, draws = orc.pixsel_libc_tables(w * h)
    dI, ab = orc.make_images(img, 3)
    o1, o2 = w * h, w * h + (w // 2) * (h // 2)
    imgs = (dI[:o1], ab[:o1], ab[o1:o2], ab[o2:])
    out = {"randomPattern": rp, "draws_mod1000": (draws % 1000).astype(np.uint16)}     # FusedWithMask only uses rand() % 1000
    out["ths"], out["thsSmoothed"] = orc.pixsel_make_hists(imgs[1], w, h)
    for pot in (3, 6):
        m, n = orc.pixsel_select(*imgs, w, h, out["thsSmoothed"], rp, pot, 1.0)
        out["select%d_idx" % pot] = np.flatnonzero(m).astype(np.int32); out["select%d_status" % pot] = m.reshape(-1)[np.flatnonzero(m)].astype(np.uint8); out["select%d_n" % pot] = n
    m, num, pot = orc.pixsel_make_maps(*imgs, w, h, rp, 600.0, 3, 1, 1.0)
    out["maps_idx"] = np.flatnonzero(m).astype(np.int32); out["maps_status"] = m.reshape(-1)[np.flatnonzero(m)].astype(np.uint8); out["maps_num_pot"] = np.array([num, pot], np.int32)
    m, num = orc.pixsel_make_maps_lidar(*imgs, w, h, rp, mask, draws, 3, 1.0)
    out["lidar_idx"] = np.flatnonzero(m).astype(np.int32); out["lidar_status"] = m.reshape(-1)[np.flatnonzero(m)].astype(np.uint8); out["lidar_num"] = np.array([num], np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_pixsel_r01.npz"), **out)
    print("wrote golden_pixsel_r01.npz with", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "golden_pixsel_r01.npz")), "bytes")


if __name__ == "__main__":
    if "--pixsel-only" in sys.argv:
        main_pixsel()
        sys.exit(0)
    if "--imm-only" not in sys.argv:
        main()
    main_imm()
    main_pixsel()
