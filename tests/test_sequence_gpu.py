"""A >= 10-keyframe synthetic sequence through the C-ABI only: track -> optimize -> marginalize points -> marginalize frame ->
prior carried into the next keyframe, GPU back-end against the CPU oracle in lock-step (tests/seq_helpers.py). After the 8th keyframe every
solveSystemF has HM != 0, frames with state != state_zero, an evolving calibration against the fixed calib_zero, and points that survived several
optimisations. Closed loop: each back-end consumes its OWN numbers; the structure (which residuals / points / frames exist) is the oracle's.

Two runs per image size:
  teacher-forced  before every keyframe the GPU back-end is handed the oracle's carried numbers (frames, calibration, HM/bM, inverse depths): what one
                  keyframe does to IDENTICAL carried state. Bar: window and tracked poses |log(T_gpu T_oracle^-1)| < 1e-5 (BASELINE.json) while every
                  residual decision agrees; a flipped borderline outlier decision (order-statistic threshold on fp32 energies) legitimately moves a
                  keyframe's poses more than rounding does: 5e-5 then. Where the window is weakly constrained the reference's own fp32 arithmetic is
                  farther than that from an all-fp64 evaluation (third back-end, same lock-step): the bar is then 1.5x that measured floor.
  closed loop     each back-end consumes its OWN numbers for the whole sequence. Two fp32 evaluations drift apart in a chaotic estimator: the strict
                  fp32 oracle against the all-fp64 oracle reaches 2e-5 .. 5e-5 over these sequences (measured, tests/seq_helpers.py dry run), which is
                  the noise floor of the reference's own arithmetic. Bar: 5e-5 per keyframe (2e-4 after a flipped decision) on the window trajectory
                  after Sim(3) alignment (the monocular gauge is held by priors only; see check()); the raw poses within max(2e-4, 3x the raw
                  distance between the two oracles at that keyframe).
The tracker's affine parameters are compared at 1e-3 (a) / 0.05 grey levels (b): b is scaled by SCALE_B = 1000 inside the LM, whose stopping
rule is |inc| < 1e-3 in scaled units."""
import numpy as np
import pytest

from helpers import pose_dist, rel_err, sim3_aligned_dist
from seq_helpers import GpuBackend, OracleBackend, SequenceDriver, make_sequence

pytestmark = pytest.mark.gpu


def check(rec, drv, tol_state, tol_clean, tol_flipped):
    fo, fg, f64 = rec["frames"]
    assert [f.fid for f in fo] == [f.fid for f in fg]
    flips = rec["state_mismatch"][1]
    tol_state["flips"] += flips
    tol = tol_clean if (tol_state["flips"] == 0 if not drv.teacher else flips == 0) else tol_flipped
    # The third back-end is the ALL-FP64 oracle run in the same lock-step: its distance from the strict fp32 oracle on this keyframe is the noise floor of
    # the reference's own arithmetic on this window. Weakly constrained windows have a high floor (the 4-frame window of the 640x480 sequence: 5.3e-5 between
    # the two oracles); the GPU has to stay within 1.5x of it there, and within the fixed bar everywhere else.
    raw = max(pose_dist(a.w2c, b.w2c) for a, b in zip(fo, fg))
    if drv.teacher:
        floor = max(pose_dist(a.w2c, b.w2c) for a, b in zip(fo, f64))
        worst = raw
    else:
        # Closed loop: a monocular window is held in place by priors only (frame 0's pose prior, depth priors, HM). Drift and flipped decisions move it
        # mostly ALONG that Sim(3) gauge (measured: two flipped residuals of 9526 at keyframe 10 of the 1224x368 run rescale the window by 8e-5, 6.6e-4 at
        # its far end, while its shape agrees to 1e-5), so the estimate itself is compared after the best Sim(3) alignment; the raw distance keeps a loose bound.
        A, Bg, B64 = [f.w2c for f in fo], [f.w2c for f in fg], [f.w2c for f in f64]
        floor, worst = sim3_aligned_dist(A, B64), sim3_aligned_dist(A, Bg)
        # the raw (gauge-included) distance is tied to what the reference's own arithmetic does in the same closed loop (VERDICT r2 #1c): 3x the raw distance
        # between the strict fp32 and the all-fp64 oracle at this keyframe, never tighter than 2e-4
        raw_floor = max(pose_dist(a.w2c, b.w2c) for a, b in zip(fo, f64))
        tol_state.setdefault("raw", []).append((raw, raw_floor))
        assert raw < max(2e-4, 3 * raw_floor), "keyframe %d: raw pose delta %.2e, raw fp64-oracle floor %.2e" % (rec["k"], raw, raw_floor)
    tol = max(tol, 1.5 * floor)
    if len(fo) < 5:                      # the first windows (2-4 frames, a few hundred points) are weakly constrained, like the toy window of test_ba_gpu.py
        tol = max(tol, 3e-5)
    tol_state.setdefault("floor", []).append(floor)
    print("  kf %2d W=%d: gpu-vs-fp32-oracle %.1e (raw %.1e), fp64-vs-fp32 oracle (floor) %.1e, flips gpu %d / fp64 %d, residuals %d" % (rec["k"], len(fo), worst, raw, floor, flips, rec["state_mismatch"][2], rec["n_res"]))
    assert worst < tol, "keyframe %d: pose delta %.2e, fp64-oracle floor %.2e (flips %d, so far %d)" % (rec["k"], worst, floor, flips, tol_state["flips"])
    for a, b in zip(fo, fg):
        if drv.teacher:                  # closed loop: each back-end's state is relative to ITS OWN linearisation point (evalPT), not comparable
            assert np.abs(a.state - b.state).max() < 1e-4 * max(1.0, np.abs(a.state).max()) + 1e-7
        assert abs(a.th - b.th) < 1e-3 * a.th
    assert rel_err(rec["calib"][1], rec["calib"][0]) < 1e-6
    ido, idg = rec["idepth"][:2]
    if not drv.teacher:                  # closed loop: the common scale of the inverse depths is the gauge (see above); compare them up to that factor
        idg = idg / np.median(idg / ido)
    assert np.median(np.abs(idg - ido) / np.abs(ido)) < 2e-5
    (Ho, bo), (Hg, bg) = rec["prior"][:2]
    assert Ho.shape == Hg.shape
    if np.abs(Ho).max() > 0:               # bM is a cancelling difference (M_b - Msc_b): its relative error is an order above H's; same floor rule as the poses
        H64, b64 = rec["prior"][2]
        # a flipped residual decision (an order statistic of fp32 energies decides: inherent) moves a marginalised point's whole contribution
        # in or out of the prior: 2e-3 of max|H| measured for ONE flip of 9897 on the 1224x368 run; without flips the teacher-forced prior agrees to 2e-4
        tight = drv.teacher and tol_state["flips"] == 0
        assert rel_err(Hg, Ho) < max(2e-4 if tight else 5e-3, 1.5 * rel_err(H64, Ho)), (rel_err(Hg, Ho), rel_err(H64, Ho), flips)
        assert rel_err(bg, bo) < max(2e-3 if tight else 5e-2, 1.5 * rel_err(b64, bo)), (rel_err(bg, bo), rel_err(b64, bo), flips)
    for (b, fid), (ok, T, aff) in rec.get("tracked", {}).items():
        if b == 1:
            ok_o, T_o, aff_o = rec["tracked"][(0, fid)]
            assert ok == ok_o
            floor_t = pose_dist(rec["tracked"][(2, fid)][1], T_o)          # the tracker's own fp32-vs-fp64 distance on this frame
            assert pose_dist(T, T_o) < max(tol, 1.5 * floor_t), "tracked frame %d: %.2e (floor %.2e)" % (fid, pose_dist(T, T_o), floor_t)
            assert abs(aff[0] - aff_o[0]) < 1e-3 and abs(aff[1] - aff_o[1]) < 0.05
    return worst


@pytest.mark.parametrize("teacher", [True, False], ids=["teacher_forced", "closed_loop"])
@pytest.mark.parametrize("w,h,n_kf", [(640, 480, 12), (1224, 368, 11)])
def test_keyframe_sequence(w, h, n_kf, teacher):
    win, kf = make_sequence(w=w, h=h, n_kf=n_kf)
    B = [OracleBackend(win), GpuBackend(win), OracleBackend(win, "f64")]
    drv = SequenceDriver(win, kf, B, teacher=teacher)
    tols = (1e-5, 5e-5) if teacher else (5e-5, 2e-4)
    tol_state = dict(flips=0)
    worst = [check(drv.bootstrap(), drv, tol_state, *tols)]
    n_marg_frames, n_marg_pts = 0, 0
    for k in range(2, n_kf):
        rec = drv.add_keyframe(k)
        worst.append(check(rec, drv, tol_state, *tols))
        n_marg_frames += len(rec["flagged"]); n_marg_pts += rec["n_marg"]
    # the sequence really exercised the carried state: frames and points were marginalised, the window reached its full size, the prior is non-zero
    assert n_marg_frames >= 3 and n_marg_pts > 200
    assert max(len(r["fids"]) for r in drv.log) == 8
    assert np.abs(drv.log[-1]["prior"][1][0]).max() > 0
    print("%dx%d %s: pose delta per keyframe:" % (w, h, "teacher-forced" if teacher else "closed loop"), ["%.1e" % x for x in worst], "flips", tol_state["flips"],
          "fp64-oracle floor:", ["%.1e" % x for x in tol_state["floor"]])
    if not teacher:
        print("  raw (gpu-vs-fp32, fp64-vs-fp32):", ["%.1e/%.1e" % x for x in tol_state["raw"]])
    if teacher:                                   # most keyframes sit at the fixed bar, whatever the floor allows on the weak ones
        assert np.median(worst) < 1e-5
    B[1].close()
