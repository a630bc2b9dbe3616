"""GPU tests at BASELINE.json's full sizes (configs[3]: 1920x1072, 5 levels, 250 k points, W=8) through size-independent properties the
domain offers, plus the edge cases of the inputs (empty / ragged / degenerate windows). The oracle is only used where it finishes in
seconds (the pyramid); everything else is checked by invariants:

  * determinism      two passes over the same window give bit-identical systems (fixed-order fp64 reductions, no float atomics)
  * linearity        the stitched systems of two disjoint point shards sum to the system of the whole window (what the multi-GPU
                     all-reduce relies on); energy and residual counts add up exactly
  * symmetry / sign  H~ symmetric to fp64 rounding, energy >= 0, count == number of active slots read back
  * descent          optimize() lowers the photometric energy and moves perturbed poses towards the truth
  * round trip       the tracker recovers a known relative pose at full resolution
"""
import dataclasses

import numpy as np
import pytest

import orc
from helpers import rel_err, pose_dist, tracker_inputs, true_rel_pose
from nalo_slam_amd import binding, synth

pytestmark = pytest.mark.gpu

FULL = dict(w=1920, h=1072, W=8, P=250000)


@pytest.fixture(scope="module")
def full_window():
    win = synth.make_window(w=FULL["w"], h=FULL["h"], W=FULL["W"], P=FULL["P"], seed=5, n_extra=1)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    return win, st6


def make_ctx(win, st6, n_slots=None):
    c = binding.Context(win.w, win.h, win.K, n_slots=n_slots or win.W + 1)
    for i in range(win.W):
        c.frame_upload(i, win.images[i])
    c.ba_set_window(list(range(win.W)), win.world_to_cam[:win.W], state6=st6)
    c.ba_set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
    c.ba_set_residuals(win.exists)
    return c


def subset(win, idx):
    return dataclasses.replace(win, host=win.host[idx], u=win.u[idx], v=win.v[idx], idepth=win.idepth[idx],
                               idepth_true=win.idepth_true[idx], color=win.color[idx], weights=win.weights[idx], exists=win.exists[idx])


def systems(c):
    E = c.ba_linearize(False)
    HA, bA = c.ba_accumulate(0)
    Hs, bs = c.ba_accumulate_sc(True)
    return E, HA, bA, Hs, bs


def test_pyramid_bit_exact_full_size(full_window):
    win, _ = full_window
    c = binding.Context(win.w, win.h, win.K, n_slots=1)
    assert c.levels == 5
    c.frame_upload(0, win.images[0])
    dI_ref, ab_ref = orc.make_images(win.images[0], c.levels)
    L = orc.lib()
    for lvl in range(c.levels):
        dI, ab = c.frame_download(0, lvl)
        o, n = L.orc_pyr_offset(win.w, win.h, lvl), (win.w >> lvl) * (win.h >> lvl)
        assert np.array_equal(dI, dI_ref[o:o + n]), "level %d texels differ" % lvl
        assert np.array_equal(ab, ab_ref[o:o + n])
    c.close()


def test_ba_determinism_symmetry_counts_full_size(full_window):
    win, st6 = full_window
    c = make_ctx(win, st6)
    E1, HA1, bA1, Hs1, bs1 = systems(c)
    st, active, _, _, _ = c.ba_get_residuals()
    nA, _, _ = c.ba_counts()
    c.close()
    c = make_ctx(win, st6)
    E2, HA2, bA2, Hs2, bs2 = systems(c)
    c.close()
    assert E1 == E2 and np.array_equal(HA1, HA2) and np.array_equal(bA1, bA2) and np.array_equal(Hs1, Hs2) and np.array_equal(bs1, bs2)
    assert E1 > 0 and np.isfinite(HA1).all() and np.isfinite(Hs1).all()
    assert np.abs(HA1 - HA1.T).max() <= 1e-12 * np.abs(HA1).max()
    # H_sc is a sum of fp32 products w*a_j*a_k whose (j,k) and (k,j) entries round differently — in the reference too (accD[j][k] and
    # accD[k][j] are separate accumulators, AccumulatedSCHessian.cpp:58-64); the solve reads the lower triangle only
    assert np.abs(Hs1 - Hs1.T).max() <= 1e-6 * np.abs(Hs1).max()
    assert nA == int(active.sum()) and nA > 0.5 * win.exists.sum()
    # the Schur complement removes information: H_A - H_sc stays positive semi-definite on the pose block (up to rounding)
    ev = np.linalg.eigvalsh(HA1 - Hs1)
    assert ev.min() > -1e-7 * ev.max()


def test_ba_shard_linearity_full_size(full_window):
    """two disjoint shards of the 250 k points: systems, energy and counts add up to the whole window's."""
    win, st6 = full_window
    c = make_ctx(win, st6)
    E, HA, bA, Hs, bs = systems(c)
    n_all = c.ba_counts()[0]
    c.close()
    acc = None
    n_sum = 0
    for r in range(2):
        part = subset(win, np.arange(len(win.host))[r::2])
        c = make_ctx(part, st6)
        got = systems(c)
        n_sum += c.ba_counts()[0]
        c.close()
        acc = got if acc is None else tuple(a + b for a, b in zip(acc, got))
    assert n_sum == n_all
    assert abs(acc[0] - E) <= 1e-9 * E
    # fp64 finishes of fp32 block partials: a different blocking changes the fp32 rounding of each partial (2e-7 relative per block)
    assert rel_err(acc[1], HA) < 2e-6 and rel_err(acc[2], bA) < 2e-5
    assert rel_err(acc[3], Hs) < 2e-6 and rel_err(acc[4], bs) < 2e-5


def test_optimize_descends_full_size(full_window):
    win, st6 = full_window
    c = make_ctx(win, st6)
    E0 = c.ba_linearize(False)
    rmse = c.ba_optimize(6, never_break=True)
    E1 = c.ba_linearize(False)
    _, w2c, _ = c.ba_get_frames()
    c.close()
    assert np.isfinite(rmse) and E1 < E0
    # gauge-free check: relative pose first -> last frame moves towards the truth
    W = win.W
    def rel(T): return synth.se3_mul(T[W - 1], synth.se3_inv(T[0]))
    d_after = pose_dist(rel(w2c), rel(win.world_to_cam[:W]))
    assert d_after < 2e-3


def test_track_round_trip_full_size(full_window):
    win, _ = full_window
    W = win.W
    c = binding.Context(win.w, win.h, win.K, n_slots=2)
    c.frame_upload(0, win.images[W - 1])
    c.frame_upload(1, win.images[W])
    Ku, Kv, idp, hdi = tracker_inputs(win, n=60000, seed=3)
    c.trk_set_ref(0, Ku, Kv, idp, hdi)
    n0 = c.trk_get_pc(0)[0].shape[0]
    assert n0 > 50000                                     # dilation grows the 60 k inputs; level 0 keeps most of them
    T_true = true_rel_pose(win, W - 1, W)
    xi = orc.se3_log(T_true) * 0.9
    ok, T, aff, lr, lf, nev = c.trk_track(1, orc.se3_exp(xi), [0, 0], [0, 0], [1, 1], c.levels - 1)
    c.close()
    assert ok and nev > 5
    assert pose_dist(T, T_true) < 2e-3
    assert abs(aff[0]) < 0.02 and abs(aff[1]) < 2.0


def test_fused_eval_many_workgroups_full_size(full_window):
    """nalo_trk_eval at 1920x1072 with 200 k and 131 073 level-0 points: 512 (the cap) and 513 -> 512 workgroups publish their partials and the one that draws the last
    ticket sums them (fourteen 16-byte coherent loads in flight per lane, rows clamped at the tail: round 4) - energy, counts and the 8x8 system against the oracle's
    evaluation of the same cloud, and bit-identical from call to call (the order of the fp64 sums is fixed, whichever workgroup comes last)"""
    win, _ = full_window
    W, w, h = win.W, win.w, win.h
    c = binding.Context(w, h, win.K, n_slots=2)
    c.frame_upload(0, win.images[W - 1]); c.frame_upload(1, win.images[W])
    dI_ref, _ = orc.make_images(win.images[W - 1], c.levels)
    dI_new, _ = orc.make_images(win.images[W], c.levels)
    T = orc.se3_exp(orc.se3_log(true_rel_pose(win, W - 1, W)) * 0.9)
    aff = np.array([0.98, 1.5], np.float32)
    d0 = win.depth[W - 1]
    rng = np.random.RandomState(2)
    for n_pts in (200000, 131073):
        idx = np.sort(rng.permutation(np.unique(rng.randint(4, h - 4, 3 * n_pts) * w + rng.randint(4, w - 4, 3 * n_pts)))[:n_pts])
        u, v = idx % w, idx // w
        ok = np.isfinite(d0[v, u]); u, v = u[ok], v[ok]
        pc = (u.astype(np.float32), v.astype(np.float32), (1.0 / d0[v, u]).astype(np.float32), win.images[W - 1][v, u].astype(np.float32))
        c.trk_set_pc(0, 0, *pc)
        trk = orc.Tracker(w, h, c.levels, win.K)
        trk.set_pc(dI_ref, 0, *pc)
        orc.lib().orc_set_sum_mode(0)
        # cutoff far above every residual: among 2e5 residuals one or two sit within an fp32 rounding of a finite cutoff, and a residual that saturates on one side only
        # moves b by its whole contribution - that decision is the small test's subject (test_tracker_gpu), the sums over many workgroups are this one's
        CUT = 1e9
        st_o = trk.calc_res(dI_new, 0, T, aff, CUT)
        H_o, b_o = trk.calc_gs(0, float(aff[0]), 0.3)
        st, H, b = c.trk_eval(1, 0, T, aff, 0.3, CUT)
        assert st[1] == st_o[1] and st[1] > 0.5 * len(u), (st[1], st_o[1])                 # same inlier count
        assert abs(st[0] - st_o[0]) / st_o[0] < 2e-6 and st[5] == st_o[5] == 0
        assert rel_err(H, H_o) < 2e-5 and rel_err(b, b_o) < 2e-5
        for _ in range(3):
            st2, H2, b2 = c.trk_eval(1, 0, T, aff, 0.3, CUT)
            assert np.array_equal(st2, st) and np.array_equal(H2, H) and np.array_equal(b2, b)
    c.close()


# ------------------------------------------------------------------------------------------------ edge cases
def test_empty_tracker_reference():
    win = synth.make_window(w=320, h=240, W=3, P=60, seed=2, n_extra=1)
    c = binding.Context(win.w, win.h, win.K, n_slots=2)
    c.frame_upload(0, win.images[2]); c.frame_upload(1, win.images[3])
    e = np.zeros(0, np.float32)
    c.trk_set_ref(0, e, e, e, e)                          # no residual on the newest keyframe: empty point clouds on every level
    for lvl in range(c.levels):
        assert c.trk_get_pc(lvl)[0].shape[0] == 0
    ok, T, aff, lr, lf, nev = c.trk_track(1, np.eye(4)[:3], [0, 0], [0, 0], [1, 1], c.levels - 1)
    assert np.isfinite(T).all() or not ok                 # the reference divides by zero terms here (NaN residuals); no fault, no hang
    c.close()


def test_degenerate_windows():
    """W = 2 (the smallest window optimize() accepts), a host with no points, a point with no residual, all residuals out of bounds."""
    win = synth.make_window(w=320, h=240, W=2, P=70, seed=4)
    c = make_ctx(win, synth.perturbed_poses(win, sigma_t=0.002, sigma_r=0.0002))
    E = c.ba_linearize(False)
    HA, bA = c.ba_accumulate(0)
    assert np.isfinite(E) and np.isfinite(HA).all() and HA.shape == (20, 20)
    assert np.isfinite(c.ba_optimize(2, never_break=True))
    c.close()

    win = synth.make_window(w=320, h=240, W=4, P=200, seed=6)
    keep = win.host != 1                                   # host frame 1 owns no point: its blocks / SC rows are empty
    part = subset(win, np.arange(len(win.host))[keep])
    ex = part.exists.copy(); ex[0, :] = 0                  # and the first point has no residual at all
    part = dataclasses.replace(part, exists=ex)
    orc.lib().orc_set_sum_mode(0)
    ba = orc.ba_from_window(part, "f32")
    c = make_ctx(part, None)
    E_o = ba.linearize_all(False); ba.apply_res()
    E = c.ba_linearize(False)
    assert abs(E - E_o) <= 1e-5 * E_o
    HA_o, bA_o = ba.accumulate(0); Hs_o, bs_o = ba.accumulate_sc(True)
    HA, bA = c.ba_accumulate(0); Hs, bs = c.ba_accumulate_sc(True)
    assert rel_err(HA, HA_o) < 2e-5 and rel_err(Hs, Hs_o) < 2e-5
    c.close()

    # every point projects outside the other frames: all slots go OOB, the systems are exactly zero, optimize() does not fault
    far = dataclasses.replace(win, idepth=np.full_like(win.idepth, 1e-6), u=np.full_like(win.u, 3.0), v=np.full_like(win.v, 3.0))
    st6 = [np.array([0.0, 0, 0, 0, 0.9, 0]) * (i % 2) for i in range(win.W)]      # alternate frames look 52 degrees away
    c = make_ctx(far, st6)
    E = c.ba_linearize(False)
    HA, bA = c.ba_accumulate(0)
    nA = c.ba_counts()[0]
    st, active, _, _, _ = c.ba_get_residuals()
    assert nA == int(active.sum())
    assert np.isfinite(E) and np.isfinite(HA).all()
    assert np.isfinite(c.ba_optimize(1, never_break=True)) or nA == 0
    c.close()


def test_two_contexts_on_two_threads():
    """the reference drives the tracker and the mapper from different threads (FullSystem::mappingLoop): two contexts used concurrently must not share
    state. Thread A tracks frames in a loop, thread B optimises a window; both must reproduce their single-threaded results bit for bit."""
    import threading
    win = synth.make_window(w=640, h=480, W=5, P=600, seed=31, n_extra=1)
    st6 = synth.perturbed_poses(win, sigma_t=0.003, sigma_r=0.0003)
    W = win.W
    Ku, Kv, idp, hdi = tracker_inputs(win, n=4000, seed=3)
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, W - 1, W)) * 0.9)

    def track_job(out, reps):
        c = binding.Context(win.w, win.h, win.K, n_slots=2)
        c.frame_upload(0, win.images[W - 1]); c.frame_upload(1, win.images[W])
        for _ in range(reps):
            c.trk_set_ref(0, Ku, Kv, idp, hdi)
            out.append(c.trk_track(1, T0, [0, 0], [0, 0], [1, 1], c.levels - 1)[1].copy())
        c.close()

    def ba_job(out, reps):
        for _ in range(reps):
            c = make_ctx(win, st6)
            c.ba_optimize(4, never_break=True)
            out.append(c.ba_get_frames()[1].copy())
            c.close()

    ref_t, ref_b = [], []
    track_job(ref_t, 1); ba_job(ref_b, 1)
    got_t, got_b = [], []
    ta, tb = threading.Thread(target=track_job, args=(got_t, 6)), threading.Thread(target=ba_job, args=(got_b, 3))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert len(got_t) == 6 and len(got_b) == 3
    assert all(np.array_equal(t, ref_t[0]) for t in got_t) and all(np.array_equal(b, ref_b[0]) for b in got_b)


def test_image_sizes_with_odd_pyramid_levels():
    """644x484 -> 322x242 -> 161x121 (odd: the pyramid stops there, 3 levels) and 1226x370 -> 613x185 (2 levels): tile edges of the fused pyramid /
    coarse-depth kernels, level tables shorter than usual, a width that is not a multiple of 32."""
    for (w, h, lv) in ((644, 484, 3), (1226, 370, 2)):
        win = synth.make_window(w=w, h=h, W=3, P=300, seed=17, n_extra=1)
        assert win.levels == lv
        c = binding.Context(win.w, win.h, win.K, n_slots=win.W + 1)
        assert c.levels == lv
        for i in range(win.W + 1):
            c.frame_upload(i, win.images[i])
        dI_ref, ab_ref = orc.make_images(win.images[2], lv)
        L = orc.lib()
        for l in range(lv):
            dI, ab = c.frame_download(2, l)
            o, n = L.orc_pyr_offset(w, h, l), (w >> l) * (h >> l)
            assert np.array_equal(dI, dI_ref[o:o + n]) and np.array_equal(ab, ab_ref[o:o + n])
        Ku, Kv, idp, hdi = tracker_inputs(win, n=2500, seed=4)
        trk = orc.Tracker(w, h, lv, win.K)
        trk.set_ref(dI_ref, Ku, Kv, idp, hdi)
        c.trk_set_ref(2, Ku, Kv, idp, hdi)
        for l in range(lv):
            a, b = c.trk_get_pc(l), trk.get_pc(l)
            assert len(a[0]) == len(b[0]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and rel_err(a[2], b[2]) < 1e-6
        T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, 2, 3)) * 0.9)
        dI_new = orc.make_images(win.images[3], lv)[0]
        ok_o, T_o = trk.track(dI_new, T0, [0, 0], [0, 0], [1, 1], lv - 1)[:2]
        ok, T = c.trk_track(3, T0, [0, 0], [0, 0], [1, 1], lv - 1)[:2]
        assert ok == ok_o and pose_dist(T, T_o) < 1e-5
        st6 = synth.perturbed_poses(win, sigma_t=0.003, sigma_r=0.0003)
        c.ba_set_window(list(range(win.W)), win.world_to_cam[:win.W], state6=st6)
        c.ba_set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
        c.ba_set_residuals(win.exists)
        ba = orc.ba_from_window(win, "f32", state6=st6)
        E_o = ba.linearize_all(False); ba.apply_res()
        E = c.ba_linearize(False)
        HA, _ = c.ba_accumulate(0); HA_o, _ = ba.accumulate(0)
        assert abs(E - E_o) <= 1e-5 * E_o and rel_err(HA, HA_o) < 2e-5
        ths, sm = c.pixsel_make_hists(2)
        ths_o, sm_o = orc.pixsel_make_hists(ab_ref[:w * h], w, h)
        assert np.array_equal(ths, ths_o) and np.array_equal(sm, sm_o)
        c.close()


def test_pyramid_six_levels_bit_exact():
    """A pyramid of PYR_LEVELS = 6 (4096x2048: the levels rule adds a level while w * h > 5000): the coarsest pixel covers 32x32 pixels of level 0, more than the
    one-pass kernel's 64x16 tile holds, so pyramid_build takes the level-by-level path - every level bit-exact against the oracle there too."""
    w, h = 4096, 2048
    rng = np.random.RandomState(11)
    img = (rng.rand(h // 8, w // 8).astype(np.float32) * 255).repeat(8, 0).repeat(8, 1) + rng.rand(h, w).astype(np.float32) * 3
    K = (0.52 * w, 0.52 * w, (w - 1) / 2.0, (h - 1) / 2.0)
    c = binding.Context(w, h, K, n_slots=1)
    assert c.levels == 6
    c.frame_upload(0, img)
    dI_ref, ab_ref = orc.make_images(img, 6)
    L = orc.lib()
    for lvl in range(6):
        dI, ab = c.frame_download(0, lvl)
        o, n = L.orc_pyr_offset(w, h, lvl), (w >> lvl) * (h >> lvl)
        assert np.array_equal(dI, dI_ref[o:o + n]), "level %d texels differ" % lvl
        assert np.array_equal(ab, ab_ref[o:o + n])
    c.close()


@pytest.mark.parametrize("w,h", [(1840, 880), (128, 5120), (1936, 1072)])
def test_pyramid_five_levels_coarse_tiles_bit_exact(w, h):
    """Five-level pyramids whose levels 3 and 4 come from the coarse workgroups of the one launch (pyr_coarse_tile, round 4): 1840x880 (level 3 = 230x110: the last coarse
    tile column and row are partial, level 4 = 115x55 is odd), 128x5120 (level 3 = 16x640: ONE coarse tile column, every tile rebuilds both flat-index neighbours of
    its rows) and 1936x1072 (level 3 = 242 wide: the last tile column holds two pixels); with the gamma table of CalibHessian::B, every level bit-exact."""
    rng = np.random.RandomState(w + h)
    img = (rng.rand(h // 16, w // 16).astype(np.float32) * 250).repeat(16, 0).repeat(16, 1) + rng.rand(h, w).astype(np.float32) * 5
    B = (255.0 * (np.arange(256) / 255.0) ** 0.8).astype(np.float32)
    K = (0.52 * w, 0.52 * w, (w - 1) / 2.0, (h - 1) / 2.0)
    c = binding.Context(w, h, K, n_slots=1)
    assert c.levels == 5
    L = orc.lib()
    tot = L.orc_pyr_offset(w, h, 5)
    for gam in (None, B):
        c.frame_upload(0, img, gammaB=gam)
        dI_ref, ab_ref = np.zeros((tot, 3), np.float32), np.zeros(tot, np.float32)
        L.orc_make_images(orc.fp(np.ascontiguousarray(img)), w, h, 5, orc.fp(gam) if gam is not None else None, orc.fp(dI_ref), orc.fp(ab_ref))
        for lvl in range(5):
            dI, ab = c.frame_download(0, lvl)
            o, n = L.orc_pyr_offset(w, h, lvl), (w >> lvl) * (h >> lvl)
            assert np.array_equal(dI, dI_ref[o:o + n]), "level %d texels differ (gamma %s)" % (lvl, gam is not None)
            assert np.array_equal(ab, ab_ref[o:o + n]), "level %d absSquaredGrad differs (gamma %s)" % (lvl, gam is not None)
    c.close()
