"""The settings of the reference that change the arithmetic of the hot path (nalo_set_settings; util/settings.cpp:71,74,128-129), through the C-ABI against
the CPU oracle on identical windows:

* a13: EnergyFunctional::calcLEnergyF_MT / calcLEnergyPt (EnergyFunctional.cpp:332-415) and calcMEnergyF (:320-329) on a window that carries a
  marginalisation prior, depth priors, idepth_zero != idepth, calib != calib_zero and frames off their linearisation point.
* setting_forceAceptStep = false: FullSystem::optimize's accept / reject branch (FullSystemOptimize.cpp:511-541) — linearise without applyRes, compare
  E + E_L + E_M, apply or loadSateBackup. The number of rejected steps must agree and the poses match as in test_ba_gpu.py.
* setting_affineOptModeA / B: 0 (no prior), < 0 (fixed: prior = setting_initialAffAPrior, JabF zeroed after the sums, Residuals.cpp:241-242) in the BA,
  and the three reduced systems of CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:1140-1162, :1243-1256) in the tracker."""
import numpy as np
import pytest

import orc
from helpers import rel_err, pose_dist, tracker_inputs, true_rel_pose
from nalo_slam_amd import binding, synth
from test_window_state_gpu import carried_inputs, realistic_prior, sub_window, make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def carried():
    win0 = synth.make_window(w=640, h=480, W=6, P=2400, seed=13)
    st6 = synth.perturbed_poses(win0, sigma_t=0.004, sigma_r=0.0004)
    has_prior0, idz0, calib_zero, aff = carried_inputs(win0)
    HM, bM, gone = realistic_prior(win0, st6, aff)
    keep = np.nonzero(gone == 0)[0]
    win = sub_window(win0, keep)
    return win, dict(st6=st6, aff=aff, has_prior=has_prior0[keep], idz=idz0[keep], calib_zero=calib_zero, HM=HM, bM=bM)


def test_l_and_m_energy(carried):
    win, x = carried
    ba, c = make_pair(win, x["st6"], x["aff"], x["has_prior"], x["idz"], x["calib_zero"], x["HM"], x["bM"])
    L_o, M_o = ba.calc_l_energy(), ba.calc_m_energy()
    L_g, M_g = c.ba_calc_l_energy(), c.ba_calc_m_energy()
    assert L_o > 0 and abs(M_o) > 0
    assert abs(L_g - L_o) < 1e-6 * abs(L_o), (L_g, L_o)
    assert abs(M_g - M_o) < 1e-9 * abs(M_o), (M_g, M_o)
    # closed form of the point part: sum deltaF^2 priorF over the points with a depth prior (setting_idepthFixPrior = 50^2)
    d = (win.idepth.astype(np.float32) - x["idz"].astype(np.float32)).astype(np.float64)
    pts = float((d * d * 2500.0 * (x["has_prior"] != 0)).sum())
    fr = ba.calc_l_energy() - pts
    assert fr > 0 and abs((L_g - pts) - fr) < 1e-6 * abs(L_o)
    c.close()


@pytest.mark.parametrize("sigma", [0.004, 0.02])
def test_optimize_with_energy_test(carried, sigma):
    """forceAcceptStep = 0: same accept / reject sequence as the oracle, same poses"""
    win, x = carried
    st6 = synth.perturbed_poses(win, sigma_t=sigma, sigma_r=sigma / 10)
    ba, c = make_pair(win, st6, x["aff"], x["has_prior"], x["idz"], x["calib_zero"], x["HM"], x["bM"])
    ba.set_settings(force_accept_step=False)
    c.set_settings(force_accept_step=False)
    r_o = ba.optimize(6)
    r_g = c.ba_optimize(6)
    its, rej = c.ba_optimize_stats()
    assert rej == ba.n_rejected(), (rej, ba.n_rejected())
    assert abs(r_g - r_o) < 1e-4 * r_o
    fr, w2c, cal = c.ba_get_frames()
    worst = max(pose_dist(w2c[i], ba.frame(i)["worldToCam"]) for i in range(win.W))
    assert worst < 3e-5, worst
    po, pg = ba.points(), c.ba_get_points()
    assert np.median(np.abs(pg["idepth"] - po["idepth"]) / np.abs(po["idepth"])) < 2e-5
    # and the default (forced) run of the same window differs from it whenever a step was rejected: the branch really ran
    ba2, _ = make_pair(win, st6, x["aff"], x["has_prior"], x["idz"], x["calib_zero"], x["HM"], x["bM"], gpu=False)
    ba2.optimize(6)
    if rej > 0:
        assert max(pose_dist(ba2.frame(i)["worldToCam"], ba.frame(i)["worldToCam"]) for i in range(win.W)) > 1e-6
    c.close()


@pytest.mark.parametrize("modes", [(0.0, 0.0), (-1.0, -1.0), (1e12, -1.0), (-1.0, 1e8)], ids=["no_prior", "fix_ab", "fix_b", "fix_a"])
def test_affine_modes_in_the_bundle_adjustment(carried, modes):
    win, x = carried
    ba, c = make_pair(win, x["st6"], x["aff"], x["has_prior"], x["idz"], x["calib_zero"], x["HM"], x["bM"])
    ba.set_settings(affine_opt_mode_a=modes[0], affine_opt_mode_b=modes[1])
    c.set_settings(affine_opt_mode_a=modes[0], affine_opt_mode_b=modes[1])
    E_o = ba.linearize_all(False); ba.apply_res()
    E_g = c.ba_linearize(False)
    assert abs(E_g - E_o) < 1e-5 * E_o
    HA_o, bA_o = ba.accumulate(0)
    HA_g, bA_g = c.ba_accumulate(0)
    assert rel_err(HA_g, HA_o) < 2e-5 and np.abs(bA_g - bA_o).max() < 5e-5 * np.abs(bA_o).max()
    HL_o, bL_o = ba.accumulate(1)
    HL_g, bL_g = c.ba_accumulate(1)
    assert np.array_equal(HL_g, HL_o) and rel_err(bL_g, bL_o) < 1e-6             # the priors themselves: exact; b_L = prior * delta (delta through fp32 states)
    # a fixed parameter leaves no gradient from the residuals in its slot (Jab_r uses the zeroed JabF) but keeps its Hessian entries
    W = win.W
    ia, ib = [4 + 8 * f + 6 for f in range(W)], [4 + 8 * f + 7 for f in range(W)]
    if modes[0] < 0 and modes[1] < 0:
        # with both zeroed, b_A's affine slots hold only what the adjoint couples in from the other frame's (also zeroed) slots: nothing
        assert np.abs(bA_g[ia]).max() == 0 and np.abs(bA_g[ib]).max() == 0
    assert np.abs(np.diag(HA_g)[ia]).min() > 0
    r_o = ba.optimize(6)
    r_g = c.ba_optimize(6)
    assert abs(r_g - r_o) < 1e-4 * r_o
    fr, w2c, cal = c.ba_get_frames()
    assert max(pose_dist(w2c[i], ba.frame(i)["worldToCam"]) for i in range(W)) < 3e-5
    st_g = np.array([np.array(f.state)[6:8] for f in fr]); st_o = np.array([ba.frame(i)["state"][6:8] for i in range(W)])
    assert np.abs(st_g - st_o).max() < 1e-5 * max(1.0, np.abs(st_o).max())
    c.close()


@pytest.mark.parametrize("modes", [(-1.0, -1.0), (1e12, -1.0), (-1.0, 1e8), (0.0, 0.0)], ids=["fix_ab", "fix_b", "fix_a", "no_prior"])
def test_affine_modes_in_the_tracker(small_window, modes):
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win, n=3000, seed=4)
    c = binding.Context(win.w, win.h, win.K, n_slots=win.W + 1)
    for i in range(win.W + 1):
        c.frame_upload(i, win.images[i])
    c.set_settings(affine_opt_mode_a=modes[0], affine_opt_mode_b=modes[1])
    c.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi)
    trk = orc.Tracker(win.w, win.h, win.levels, win.K)
    trk.set_affine_modes(*modes)
    dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels)
    dI_new, _ = orc.make_images(win.images[win.W], win.levels)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.9)
    ok_g, T_g, aff_g = c.trk_track(win.W, T0, [0.02, 3.0], [0, 0], [1, 1], c.levels - 1)[:3]
    ok_o, T_o, aff_o = trk.track(dI_new, T0, [0.02, 3.0], [0, 0], [1, 1], win.levels - 1)[:3]
    assert ok_g == ok_o and ok_o
    assert pose_dist(T_g, T_o) < 1e-5
    assert abs(aff_g[0] - aff_o[0]) < 1e-3 and abs(aff_g[1] - aff_o[1]) < 0.05
    if modes[0] < 0:
        assert aff_g[0] == 0 and aff_o[0] == 0
    if modes[1] < 0:
        assert aff_g[1] == 0 and aff_o[1] == 0
    c.close()
