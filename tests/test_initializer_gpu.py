"""SURVEY 8(f) rank 2, the whole two-frame initialiser through the C-ABI (nalo_init_set_first / nalo_init_track_frame) against the CPU oracle
(oracle/orc_initfull.c) on a synthetic forward-moving sequence: CoarseInitializer::setFirst (point selection of every level, makeNN) and ::trackFrame
(propagateDown, resetPoints, the LM loop around calcResAndGS / doStep, calcEC, applyStep, optReg, propagateUp) — reference
src/FullSystem/CoarseInitializer.cpp:81-285, 634-1069.

setFirst is integer / exact work: the selected points, their order, my_type, the 10 neighbours (nanoflann's tie order; the oracle's tree is itself pinned
against the reference's nanoflann.h in tests/test_oracle_cpu.py), parents and the fp32 neighbour weights must be EQUAL.
trackFrame is an LM loop with accept / reject decisions on fp32 energies summed in a different order on the device: poses are compared at 1e-5
(BASELINE.json's pose bar) per frame while both sides take the same decisions (same number of evaluations); a run whose decision sequences diverge is
still required to agree on snapped / the return value and to stay within 1e-3 (it is then two valid LM paths). The per-point state (idepth, iR,
lastHessian) comes out of ~50 chained 1-D Gauss-Newton updates per point, some of them on nearly flat energies: it is compared through the ALL-FP64
oracle run beside the two — the GPU must be as close to it as the strict fp32 oracle is (median and 99 % quantile, x2), with a loose fixed backstop."""
import numpy as np
import pytest

import orc
from helpers import pose_dist
from nalo_slam_amd import binding, synth

pytestmark = pytest.mark.gpu


def run_pair(w, h, n_frames, step_z=0.15):
    win = synth.make_window(w=w, h=h, W=2, P=20, seed=3, n_extra=n_frames - 1, step_z=step_z, yaw_deg=0.1)
    rp, _ = orc.pixsel_libc_tables(w * h)
    ini = orc.Initializer(w, h, win.levels, win.K)
    sf_o = ini.set_first(win.images[0], rp)
    c = binding.Context(w, h, win.K, n_slots=n_frames + 1)
    for i in range(n_frames + 1):
        c.frame_upload(i, win.images[i])
    c.pixsel_set_random(rp)
    num, sf_g = c.init_set_first(0)
    return win, ini, c, (sf_o, sf_g, num)


@pytest.mark.parametrize("w,h", [(640, 480), (1224, 368)])
def test_set_first_points_and_neighbours_equal(w, h):
    win, ini, c, (sf_o, sf_g, num) = run_pair(w, h, 1)
    assert sf_o == sf_g
    assert [ini.num(l) for l in range(win.levels)] == list(num)
    assert num[0] > 1000
    for l in range(win.levels):
        g = c.init_points(l)
        for k in ("u", "v", "idepth", "iR", "isGood", "my_type", "outlierTH", "parent", "neighbours", "neighboursDist", "parentDist"):
            assert np.array_equal(g[k], ini.get(l, k)), (l, k)
    c.close()


@pytest.mark.parametrize("w,h,n_frames", [(640, 480, 9), (1224, 368, 8)])
def test_track_frames_match_oracle(w, h, n_frames):
    win, ini, c, _ = run_pair(w, h, n_frames)
    rp, _ = orc.pixsel_libc_tables(w * h)
    ini64 = orc.Initializer(w, h, win.levels, win.K, "f64")
    ini64.set_first(win.images[0], rp)
    same_path, first_split = True, None
    returned = []
    for i in range(1, n_frames + 1):
        ok_o = ini.track_frame(win.images[i])
        ok_g = c.init_track_frame(i)
        ini64.track_frame(win.images[i])
        so, sg, s64 = ini.state(), c.init_state(), ini64.state()
        assert (so["snapped"], so["frameID"], so["snappedAt"]) == (sg["snapped"], sg["frameID"], sg["snappedAt"]), i
        assert ok_o == ok_g
        returned.append(ok_g)
        if same_path and so["n_evals"] != sg["n_evals"]:
            first_split = i
        same_path = same_path and so["n_evals"] == sg["n_evals"]
        d = pose_dist(sg["thisToNext"], so["thisToNext"])
        # the LM step solves Hl = H - Hsc/(1+lambda) in FLOAT (Mat88f, :182-197): the subtraction cancels, so two fp32 evaluations of the same path differ by
        # more than the 1e-5 bar on some frames; the all-fp64 oracle run beside them measures that floor (x1.5) when it took the same decisions
        # ... and the loop is chaotic: +1 ulp on 1 % of the new frames' pixels moves the ORACLE's own result by 1.2e-4 at frame 2 and changes its decision
        # path (tests/test_initializer_cpu.py::test_oracle_initializer_sensitivity). 2e-4 on an identical decision path is therefore the bar here; the
        # first frame (short path, 33 evaluations) keeps the 1e-5 bar.
        floor = pose_dist(s64["thisToNext"], so["thisToNext"]) if so["n_evals"] == s64["n_evals"] else 0.0
        bar = max(1e-5, 1.5 * floor) if i == 1 else 2e-4
        assert d < (bar if same_path else 1e-3), (i, d, floor, so["n_evals"], sg["n_evals"], s64["n_evals"])
        assert np.abs(sg["aff"] - so["aff"]).max() < 1e-6
        if same_path:
            for l in range(win.levels):
                g = c.init_points(l)
                good_o = ini.get(l, "isGood")
                assert (g["isGood"] != good_o).mean() < 2e-3
                m = (g["isGood"] == 1) & (good_o == 1)
                m64 = m & (ini64.get(l, "isGood") == 1)
                for k in ("idepth", "iR", "lastHessian"):
                    a, b, t = g[k][m64], ini.get(l, k)[m64], ini64.get(l, k)[m64]
                    sc = np.maximum(np.abs(t), 1e-3)
                    mine, ref = np.abs(a - t) / sc, np.abs(b - t) / sc
                    if i == 1 and so["n_evals"] == s64["n_evals"]:       # first frame, same decisions in fp64: its distance from the fp32 oracle is the floor (later frames: chaotic, see above)
                        for q in (0.5, 0.99):
                            assert np.quantile(mine, q) < 2 * np.quantile(ref, q) + 1e-6, (i, l, k, q, np.quantile(mine, q), np.quantile(ref, q))
                    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
                    assert np.median(d) < (2e-3 if i == 1 else 2e-2) and (i > 1 or np.quantile(d, 0.99) < 0.1), (i, l, k, np.median(d), np.quantile(d, 0.99))
    # identical decisions are required on the first three frames; beyond that the two fp32 evaluations may part ways (the oracle itself does under a 1-ulp
    # input perturbation, tests/test_initializer_cpu.py) — snapped / frameID / the return value were still required to agree on every frame above
    assert first_split is None or first_split > 3, "the device LM left the oracle's accept/reject path at frame %s" % first_split
    assert returned[-1] and not returned[0]          # the sequence is long enough for `snapped && frameID > snappedAt + 5`
    # the recovered translation direction is the true one (scale is free in the initialiser)
    t = c.init_state()["thisToNext"][:, 3]
    Tt = synth.se3_mul(win.world_to_cam[n_frames], synth.se3_inv(win.world_to_cam[0]))[:, 3]
    assert np.dot(t, Tt) / (np.linalg.norm(t) * np.linalg.norm(Tt)) > 0.95
    c.close()
