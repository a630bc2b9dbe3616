"""CPU tests of the oracle itself (parity unpinned by the reference: these are the pins we do have).
 - SE(3): the element list of the reference's vendored Sophus test (thirdparty/Sophus/sophus/test_se3.cpp:41-60) as KATs
   for exp/log/Adj/group identities;
 - oracle-free cross-checks: finite differences of the residual Jacobians, the Schur identity of accumulate+SC+stitch
   against a dense J^T J assembled in numpy, LM recovery of a known synthetic pose, BA convergence;
 - reference-order fp32 accumulators (SSE lanes + 3-tier shiftUp) stay inside a stated band of the fp64 sums."""
import numpy as np
import pytest

import orc
from helpers import rel_err, tracker_inputs, true_rel_pose, pose_dist
from nalo_slam_amd import synth


def _so3(w):
    return synth.so3_exp(np.asarray(w, float))


def _se3(R, t):
    return np.concatenate([R, np.asarray(t, float)[:, None]], axis=1)


def sophus_elements():
    I = np.eye(3)
    e = [
        _se3(_so3([0.2, 0.5, 0.0]), [0, 0, 0]),
        _se3(_so3([0.2, 0.5, -1.0]), [10, 0, 0]),
        _se3(_so3([0.0, 0.0, 0.0]), [0, 100, 5]),
        _se3(_so3([0.0, 0.0, 0.00001]), [0, 0, 0]),
        _se3(_so3([0.0, 0.0, 0.00001]), [0, -0.00000001, 0.0000000001]),
        _se3(_so3([0.0, 0.0, 0.00001]), [0.01, 0, 0]),
        _se3(_so3([np.pi, 0, 0]), [4, -5, 0]),
    ]
    e.append(synth.se3_mul(synth.se3_mul(_se3(_so3([0.2, 0.5, 0.0]), [0, 0, 0]), _se3(_so3([np.pi, 0, 0]), [0, 0, 0])),
                           _se3(_so3([-0.2, -0.5, -0.0]), [0, 0, 0])))
    e.append(synth.se3_mul(synth.se3_mul(_se3(_so3([0.3, 0.5, 0.1]), [2, 0, -7]), _se3(_so3([np.pi, 0, 0]), [0, 0, 0])),
                           _se3(_so3([-0.3, -0.5, -0.1]), [0, 6, 0])))
    return e


def test_se3_sophus_kats():
    for T in sophus_elements():
        xi = orc.se3_log(T)
        T2 = orc.se3_exp(xi)
        assert np.abs(T2 - T).max() < 1e-9 * max(1.0, np.abs(T).max())
        Ti = orc.se3_inv(T)
        assert np.abs(orc.se3_mul(T, Ti) - _se3(np.eye(3), [0, 0, 0])).max() < 1e-9
        # Adj(T) x = vee(T hat(x) T^-1): check through exp on a small tangent
        x = np.array([1e-4, -2e-4, 3e-4, 2e-4, 1e-4, -3e-4])
        lhs = orc.se3_log(orc.se3_mul(orc.se3_mul(T, orc.se3_exp(x)), Ti))
        assert np.abs(lhs - orc.se3_adj(T) @ x).max() < 1e-6 * max(1.0, np.abs(orc.se3_adj(T)).max())
    for xi in ([0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 0, 1, 0.1, 0, 0], [0, -5, 10, 0, 0, 0], [-1, 1, 0, 0, 0, 1], [20, -1, 0, -1, 1, 0],
               [30, 5, -1, 20, -1, 0]):
        xi = np.array(xi, float)
        T = orc.se3_exp(xi)
        R = T[:, :3]
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
        if np.linalg.norm(xi[3:]) < np.pi:
            assert np.abs(orc.se3_log(T) - xi).max() < 1e-9 * max(1, np.abs(xi).max())


def test_ldlt_against_numpy():
    rng = np.random.RandomState(0)
    L = orc.lib()
    for n in (8, 36, 68):
        A = rng.randn(n, n)
        A = A @ A.T + 1e-3 * np.eye(n)
        S = np.diag(10.0 ** rng.uniform(-3, 3, n))
        A = S @ A @ S
        b, x = rng.randn(n), np.zeros(n)
        L.orc_ldlt_solve(n, orc.dp(np.ascontiguousarray(A).reshape(-1)), orc.dp(b), orc.dp(x))
        assert np.abs(x - np.linalg.solve(A, b)).max() <= 1e-9 * np.abs(x).max()


def test_pyramid_properties(small_window):
    win = small_window
    dI, ab = orc.make_images(win.images[0], win.levels)
    L = orc.lib()
    assert win.levels == L.orc_pyr_levels(win.w, win.h) == 4
    I0 = dI[:win.w * win.h, 0].reshape(win.h, win.w)
    assert np.array_equal(I0, win.images[0])
    o1 = L.orc_pyr_offset(win.w, win.h, 1)
    I1 = dI[o1:o1 + (win.w // 2) * (win.h // 2), 0].reshape(win.h // 2, win.w // 2)
    box = 0.25 * (I0[0::2, 0::2] + I0[0::2, 1::2] + I0[1::2, 0::2] + I0[1::2, 1::2])
    assert np.abs(I1 - box).max() < 1e-4
    dx = dI[:win.w * win.h, 1].reshape(win.h, win.w)
    assert np.abs(dx[5, 5] - 0.5 * (I0[5, 6] - I0[5, 4])) < 1e-6
    assert np.all(dx[0] == 0) and np.all(dx[-1] == 0)                       # rows the reference leaves uninitialised
    assert abs(dx[3, 0] - 0.5 * (I0[3, 1] - I0[2, -1])) < 1e-6              # flat-index wrap at x = 0


def _tracker(win, kind="f32"):
    Ku, Kv, nid, hdi = tracker_inputs(win)
    trk = orc.Tracker(win.w, win.h, win.levels, win.K, kind)
    dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels, kind)
    dI_new, _ = orc.make_images(win.images[win.W], win.levels, kind)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    return trk, dI_new


def test_tracker_gradient_matches_finite_differences(small_window):
    """b (Vec8) is the gradient of the mean energy w.r.t. the scaled left increment: compare with central differences of
    calcRes' energy (Huber inactive region dominates; image gradients are interpolated, not exact): 10% + absolute floor."""
    win = small_window
    trk, dI_new = _tracker(win)
    T = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.97)
    aff = np.array([1.0, 0.0], np.float32)
    lvl = 1
    st = trk.calc_res(dI_new, lvl, T, aff, 1e9)
    H, b = trk.calc_gs(lvl, 1.0, 0.0)
    n = st[1]
    npad = (int(n) + 3) // 4 * 4
    sc = np.array([1, 1, 1, 0.5, 0.5, 0.5, 10, 1000.0])
    g = np.zeros(6)
    for k in range(6):
        eps = 2e-4
        d = np.zeros(6)
        d[k] = eps
        Ep = trk.calc_res(dI_new, lvl, synth.se3_mul(orc.se3_exp(d * sc[k]), T), aff, 1e9)
        Em = trk.calc_res(dI_new, lvl, synth.se3_mul(orc.se3_exp(-d * sc[k]), T), aff, 1e9)
        if Ep[1] != n or Em[1] != n:
            g[k] = np.nan
            continue
        g[k] = (Ep[0] - Em[0]) / (2 * eps) / (2 * npad)
    ok = np.isfinite(g)
    assert ok.sum() >= 4
    scale = np.abs(b[:6]).max()
    assert np.abs(g[ok] - b[:6][ok]).max() < 0.1 * scale
    assert np.all(np.linalg.eigvalsh(H) > -1e-9 * np.abs(H).max())          # PSD


def test_tracker_lm_recovers_known_pose(small_window):
    win = small_window
    trk, dI_new = _tracker(win)
    Ttrue = true_rel_pose(win, win.W - 1, win.W)
    ok, T, aff, lr, lf = trk.track(dI_new, orc.se3_exp(orc.se3_log(Ttrue) * 0.8), [0, 0], [0, 0], [1, 1], win.levels - 1)
    assert ok == 1 and pose_dist(T, Ttrue) < 2e-3
    assert lr[0] < lr[win.levels - 1] * 2 and np.isnan(lr[4])


def test_tracker_reference_order_band(small_window):
    """SSE-lane + 3-tier fp32 accumulation (the reference's order) vs plain fp64 sums: H within 2e-5 of max|H|."""
    win = small_window
    trk, dI_new = _tracker(win)
    T = true_rel_pose(win, win.W - 1, win.W)
    aff = np.array([1.0, 0.0], np.float32)
    L = orc.lib()
    L.orc_set_sum_mode(0)
    s0 = trk.calc_res(dI_new, 0, T, aff, 20.0)
    H0, b0 = trk.calc_gs(0, 1.0, 0.0)
    L.orc_set_sum_mode(1)
    s1 = trk.calc_res(dI_new, 0, T, aff, 20.0)
    H1, b1 = trk.calc_gs(0, 1.0, 0.0)
    L.orc_set_sum_mode(0)
    assert s0[1] == s1[1] and abs(s0[0] - s1[0]) < 1e-5 * s0[0]
    assert rel_err(H1, H0) < 2e-5 and rel_err(b1, b0) < 1e-4


def _dense_system(win, ba):
    """numpy J^T J over all active residuals, in the global parametrisation [calib(4) | frames(8 each) | idepths]."""
    W, P, n = win.W, len(win.host), 8 * win.W + 4
    st, ac, jp, en = ba.slots()
    adH, adT, _ = ba.adjoints()
    rows, rhs = [], []
    for p in range(P):
        for t in range(W):
            if not ac[p, t]:
                continue
            J = ba.residual(p, t)["J"]
            resF, Jx, Jy, Cx, Cy, dd = J[0:8], J[8:14], J[14:20], J[20:24], J[24:28], J[28:30]
            JI0, JI1, Ja, Jb = J[30:38], J[38:46], J[46:54], J[54:62]
            h = win.host[p]
            k = h + t * W
            for i in range(8):
                loc = np.concatenate([JI0[i] * Jx + JI1[i] * Jy, [Ja[i], Jb[i]]])
                row = np.zeros(n + P)
                row[0:4] = JI0[i] * Cx + JI1[i] * Cy
                row[4 + 8 * h:12 + 8 * h] += adH[k] @ loc
                row[4 + 8 * t:12 + 8 * t] += adT[k] @ loc
                row[n + p] = JI0[i] * dd[0] + JI1[i] * dd[1]
                rows.append(row)
                rhs.append(resF[i])
    Jf, rf = np.array(rows), np.array(rhs)
    return Jf.T @ Jf, Jf.T @ rf


@pytest.mark.parametrize("kind", ["f32", "f64"])
def test_ba_schur_identity(kind):
    """accumulate(A) + accumulateSC + both stitches == dense J^T J with the idepths eliminated (oracle-free check)."""
    win = synth.make_window(w=320, h=240, W=3, P=90, seed=3)
    st6 = synth.perturbed_poses(win, sigma_t=0.003, sigma_r=0.0003)
    orc.lib(kind).orc_set_sum_mode(0)
    ba = orc.ba_from_window(win, kind, state6=st6)
    ba.linearize_all(False)
    ba.apply_res()
    HA, bA = ba.accumulate(0)
    Hs, bs = ba.accumulate_sc(True)
    n = 8 * win.W + 4
    Hf, bf = _dense_system(win, ba)
    Hdd = np.diag(Hf[n:, n:]).copy()
    Hdi = np.where(Hdd > 0, 1.0 / np.maximum(Hdd, 1e-10), 0)
    Hpd = Hf[:n, n:]
    tol = 1e-6 if kind == "f32" else 1e-7        # EFPoint fields (HdiF, Hcd...) are float in both builds
    assert rel_err(HA, Hf[:n, :n]) < tol and rel_err(bA, bf[:n]) < tol
    assert rel_err(Hs, (Hpd * Hdi) @ Hpd.T) < tol and rel_err(bs, (Hpd * Hdi) @ bf[n:]) < 10 * tol
    assert np.abs(HA - HA.T).max() <= 1e-12 * np.abs(HA).max()


def test_residual_jacobian_finite_differences():
    """d resF / d idepth and d resF / d (target pose) against central differences of the residual itself."""
    # band-limited texture (freq_scale): the reference's J uses interpolated central differences, which only equal the
    # derivative of the bilinear interpolant on smooth images
    win = synth.make_window(w=320, h=240, W=3, P=60, seed=4, freq_scale=0.2, min_grad2=4.0)
    kind = "f64"
    ba = orc.ba_from_window(win, kind)
    ba.linearize_all(False)
    ba.apply_res()
    st, ac, _, _ = ba.slots()
    checked = 0
    fds, ans = [], []
    for p in range(len(win.host)):
        for t in range(win.W):
            if not ac[p, t] or checked >= 12:
                continue
            r0 = ba.residual(p, t)["J"]
            resF, dd, JI0, JI1, hw0 = r0[0:8], r0[28:30], r0[30:38], r0[38:46], r0[54:62]
            if np.abs(resF).max() > 6:              # keep away from the Huber kink
                continue
            eps = 1e-4 * win.idepth[p]
            out = []
            for s in (+1, -1):
                idp = win.idepth.copy()
                idp[p] += s * eps
                b2 = orc.ba_from_window(win, kind)
                b2.L.orc_ba_set_idepth(b2.h_, orc.fp(idp))
                b2.prepare()
                b2.linearize_all(False)
                b2.apply_res()
                Jp = b2.residual(p, t)["J"]
                out.append(Jp[0:8] / Jp[54:62])         # un-weighted residual: resF / hw (JabF[1] = hw)
            fd = (out[0] - out[1]) / (2 * eps)
            an = (JI0 * dd[0] + JI1 * dd[1]) / hw0      # the weights' own derivative is not part of the reference's J
            fds.append(fd)
            ans.append(an)
            checked += 1
    assert checked >= 6
    # central-difference image gradients vs the exact slope of the bilinear interpolant: equal up to the texture's curvature,
    # so the check is statistical: slope 1 +- 10 %, correlation > 0.98 over all (residual, pattern pixel) pairs
    fds, ans = np.concatenate(fds), np.concatenate(ans)
    slope = (fds @ ans) / (ans @ ans)
    assert abs(slope - 1) < 0.1 and np.corrcoef(fds, ans)[0, 1] > 0.98


def test_ba_optimize_converges_and_reference_order_band():
    win = synth.make_window(w=640, h=480, W=4, P=400, seed=7)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    res = {}
    for name, kind, mode in (("f32", "f32", 0), ("ref", "f32", 1), ("f64", "f64", 0)):
        orc.lib(kind).orc_set_sum_mode(mode)
        ba = orc.ba_from_window(win, kind, state6=st6)
        before = max(pose_dist(ba.frame(f)["worldToCam"], win.world_to_cam[f]) for f in range(win.W))
        ba.optimize(6)
        res[name] = [ba.frame(f)["worldToCam"] for f in range(win.W)]
        orc.lib(kind).orc_set_sum_mode(0)
        after = max(pose_dist(res[name][f], win.world_to_cam[f]) for f in range(win.W))
        assert after < 0.5 * before
    # the reference's own fp32 summation order moves the poses by ~1e-5: that is the noise floor of "the reference"
    d_ref = max(pose_dist(res["ref"][f], res["f32"][f]) for f in range(win.W))
    d_64 = max(pose_dist(res["f64"][f], res["f32"][f]) for f in range(win.W))
    assert d_ref < 1e-4 and d_64 < 1e-5


def test_dense_make_map_oracle():
    L = orc.lib()
    w, h = 160, 120
    rng = np.random.RandomState(0)
    mask = np.zeros((h, w), np.float32)
    mask[30:90, 40:130] = 7.0
    mask[35:40, 50:60] = 3.0
    dI = rng.rand(h * w, 3).astype(np.float32)
    bgr = rng.randint(0, 255, (h * w, 3)).astype(np.uint8)
    rect = np.zeros(4, np.int32)
    L.orc_dense_bbox(orc.fp(mask), w, h, 7.0, orc.ip(rect))
    assert list(rect) == [40, 129, 30, 89]
    plane = np.array([0.0, 1.0, 0.0, -1.6], np.float32)
    fx = fy = 100.0
    cx, cy = 79.5, 59.5
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    cap = w * h
    ou, ov = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    oid, oc, ob = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
    acc = np.zeros(1, np.int32)
    n = L.orc_dense_make_map(orc.fp(mask), orc.fp(dI), orc.u8p(bgr), w, h, orc.fp(plane), 7.0, orc.ip(rect), 1 / fx, 1 / fy, cx, cy,
                             orc.dp(c2w), orc.ip(ou), orc.ip(ov), orc.fp(oid), orc.fp(oc), orc.u8p(ob), orc.ip(acc))
    sel = [(i, j) for i in range(30, 89) for j in range(40, 129) if mask[i, j] == 7.0 and (i % 3 == 0 or j % 3 == 0)]
    assert n == len(sel) and (ov[0], ou[0]) == sel[0] and (ov[n - 1], ou[n - 1]) == sel[-1]
    i, j = sel[10]
    depth = 1.6 / ((i - cy) / fy)
    assert abs(1 / oid[10] - depth) < 1e-4 * abs(depth)


def test_immature_points_filter_and_activation_recover_true_depth():
    """SURVEY 8(f) rank 1, oracle-free check of the restatement: tracing an immature point along the epipolar line of later frames brackets
    the scene's true inverse depth, and the 1-D Gauss-Newton of the activation reduces a 5 % depth error."""
    from imm_helpers import imm_points, host_to_new, true_idepth
    win = synth.make_window(w=320, h=240, W=4, P=100, seed=9, n_extra=2, step_z=0.25, yaw_deg=0.4)
    W = win.W
    dI = [orc.make_images(win.images[i], 1)[0] for i in range(W + 2)]
    u, v, host = imm_points(win, per_host=400, seed=2)
    n = len(u)
    color, weights, gradH, eth = [np.zeros((n, k), np.float32) for k in (8, 8, 3)] + [np.zeros(n, np.float32)]
    for h in range(W):
        m = host == h
        color[m], weights[m], gradH[m], eth[m] = orc.imm_create(dI[h], win.w, win.h, u[m], v[m])
    assert np.isfinite(eth).all() and np.all(eth == 8 * 144.0)
    uf, vf = u.astype(np.float32), v.astype(np.float32)
    st = [np.zeros(n, np.float32), np.full(n, np.nan, np.float32), np.full(n, 5, np.int32), np.full(n, 10000, np.float32)]
    for new in (W, W + 1):
        KRKi, Kt, aff = host_to_new(win, new)
        st = list(orc.imm_trace(dI[new], win.w, win.h, uf, vf, color, weights, gradH, eth, host, KRKi, Kt, aff, *st)[:4])
    idmin, idmax, status, quality = st
    good = (status == 0) & (quality > 3)                  # setting_minTraceQuality = 3: what the activation logic trusts (settings.cpp:166)
    idt = true_idepth(win, u, v, host)
    assert good.sum() > 0.25 * n
    inside = (idt[good] > idmin[good] * 0.9) & (idt[good] < idmax[good] * 1.1)
    assert inside.mean() > 0.9
    assert (idmax[good] >= idmin[good]).all()
    # activation from a 5 % wrong interval midpoint
    rng = np.random.RandomState(4)
    mid = idt * (1 + 0.05 * rng.randn(n)).astype(np.float32)
    st6 = synth.perturbed_poses(win, sigma_t=0.0, sigma_r=0.0)
    ba = orc.ba_from_window(win, "f32", state6=st6)
    Rt, af = ba.precalc_rt()
    res, idp, rin = orc.imm_optimize(dI[:W], win.w, win.h, win.K, Rt, af, host, uf, vf, color, weights, eth, mid * 0.9, mid * 1.1, 1)
    act = res == 1
    assert act.sum() > 0.5 * n and (rin[act].sum(1) >= 1).all() and (rin[~act] == 0).all()
    err0 = np.abs(mid[act] - idt[act]) / idt[act]
    err1 = np.abs(idp[act] - idt[act]) / idt[act]
    assert np.median(err1) < 0.6 * np.median(err0)         # small image, short baselines: three damped GN steps halve the error


def test_initializer_pass_schur_identity_and_descent():
    """SURVEY 8(f) rank 2, oracle-free checks of the calcResAndGS restatement: (i) the Schur accumulator equals sum_i w_i Jb_i Jb_i^T rebuilt in
    numpy from the returned per-point buffers, (ii) H is the Gauss-Newton matrix of b: a damped step from a 10 % wrong pose lowers the energy,
    (iii) the reference's E.num = 2 npts and alphaEnergy = min(alphaW |t|^2 n, alphaK n) behaviour."""
    win = synth.make_window(w=320, h=240, W=2, P=50, seed=13, n_extra=0, step_z=0.12, yaw_deg=0.3)
    dI0 = orc.make_images(win.images[0], 1)[0]
    dI1 = orc.make_images(win.images[1], 1)[0]
    rng = np.random.RandomState(5)
    n = 1500
    u = rng.randint(6, win.w - 7, n).astype(np.float32); v = rng.randint(6, win.h - 7, n).astype(np.float32)
    d = win.depth[0][v.astype(int), u.astype(int)]
    idt = np.where(np.isfinite(d), 1.0 / d, 0.1).astype(np.float32)
    pts = dict(u=u, v=v, idepth_new=idt, iR=idt, isGood=np.ones(n, np.uint8), energy=np.zeros((n, 2), np.float32), outlierTH=np.full(n, 8 * 144.0, np.float32),
               lastHessian_new=np.zeros(n, np.float32), Jb=np.zeros((n, 10), np.float32))
    K4 = np.array(win.K, np.float64)
    T_true = synth.se3_mul(win.world_to_cam[1], synth.se3_inv(win.world_to_cam[0]))
    T0 = orc.se3_exp(orc.se3_log(T_true) * 0.9)
    r0 = orc.init_calc_res_and_gs(dI0, dI1, win.w, win.h, K4, T0, (0.0, 0.0), pts, alphaK=1e9, alphaW=0.0)
    g = r0["isGood_new"] == 1
    assert g.sum() > 0.8 * n and r0["E3"][2] == 2 * n and r0["E3"][1] == 0.0
    Jb = r0["Jb"][g].astype(np.float64)
    Hs = (Jb[:, :9] * Jb[:, 9:10]).T @ Jb[:, :9]
    assert np.abs(Hs[:8, :8] - r0["Hsc"]).max() < 1e-5 * np.abs(r0["Hsc"]).max() and np.abs(Hs[:8, 8] - r0["bsc"]).max() < 1e-5 * np.abs(r0["bsc"]).max()
    Hr, br = r0["H"] - r0["Hsc"], r0["b"] - r0["bsc"]
    inc = -np.linalg.solve(Hr + 0.1 * np.diag(np.diag(Hr)), br)
    r1 = orc.init_calc_res_and_gs(dI0, dI1, win.w, win.h, K4, synth.se3_mul(orc.se3_exp(inc[:6]), T0), (float(inc[6]), float(inc[7])), pts, alphaK=1e9, alphaW=0.0)
    rt = orc.init_calc_res_and_gs(dI0, dI1, win.w, win.h, K4, T_true, (0.0, 0.0), pts, alphaK=1e9, alphaW=0.0)
    assert r1["E3"][0] < r0["E3"][0] and rt["E3"][0] < r0["E3"][0]          # the photometric minimum sits within noise of the true pose
    ra = orc.init_calc_res_and_gs(dI0, dI1, win.w, win.h, K4, T0, (0.0, 0.0), pts)                 # reference constants: alphaW = 150^2, alphaK = 2.5^2
    tsq = float((T0[:, 3] ** 2).sum())
    assert abs(ra["E3"][1] - min(150.0 * 150.0 * tsq * n, 2.5 * 2.5 * n)) < 1e-3 * ra["E3"][1]


def test_distance_map_matches_dilation_formulation():
    """the queue-based BFS of CoarseDistanceMap::growDistBFS restated as array dilations in numpy (independent formulation):
    level k = unreached pixels with a level-(k-1) neighbour that is not on the image border, 8-neighbourhood for odd k, 4 for even k."""
    rng = np.random.RandomState(3)
    w1, h1, n = 97, 61, 40
    u = rng.randint(1, w1, n).astype(np.float32); v = rng.randint(1, h1, n).astype(np.float32)
    I = np.tile(np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32), (2, 1)); Z = np.zeros((2, 3), np.float32)
    got = orc.dist_make_map(w1, h1, 1, np.zeros(n, np.int32), u, v, np.ones(n, np.float32), I, Z)      # identity projection: seeds at (u, v)
    ref = np.full((h1, w1), 1000.0, np.float32)
    ref[v.astype(int), u.astype(int)] = 0
    inner = np.zeros((h1, w1), bool); inner[1:-1, 1:-1] = True
    for k in range(1, 40):
        src = (ref == k - 1) & inner
        grow = np.zeros_like(src)
        shifts = [(0, 1), (0, -1), (1, 0), (-1, 0)] + ([(1, 1), (1, -1), (-1, 1), (-1, -1)] if k % 2 == 1 else [])
        for dy, dx in shifts:
            grow |= np.roll(np.roll(src, dy, 0), dx, 1)          # `inner` keeps the wrap-around of roll out of play
        ref[grow & (ref > k)] = k
    assert np.array_equal(got, ref)


def test_pixel_selector_matches_closed_form_and_bookkeeping():
    """PixelSelector::select restated WITHOUT the reference's -2 flags (independent formulation, the one the HIP kernels use): every
    pixel is a candidate of all three levels; a 2pot block keeps its level-2 pixel iff none of its cells selected, a 4pot block its level-3 pixel iff nothing
    below selected; maxima are first-in-scan-order. Plus the bookkeeping of makeMaps (sub-selection) and FusedWithMask (counts = final statuses)."""
    w, h, pot, thf = 160, 128, 3, np.float32(1.0)
    win = synth.make_window(w=w, h=h, W=2, P=20, seed=4, n_extra=0)
    img = win.images[1].copy()
    xs = np.arange(40, 100)
    img[30:80, xs] = (100 + 60 * ((xs // 3) % 2)).astype(np.float32)[None, :]           # dy == 0 exactly: cells whose direction is (0,1) select nothing
    dI, ab = orc.make_images(img, 3)
    o1, o2 = w * h, w * h + (w // 2) * (h // 2)
    imgs = (dI[:o1], ab[:o1], ab[o1:o2], ab[o2:])
    rp, draws = orc.pixsel_libc_tables(w * h)
    _, sm = orc.pixsel_make_hists(imgs[1], w, h)
    got, n = orc.pixsel_select(*imgs, w, h, sm, rp, pot, float(thf))
    smp = np.zeros(len(sm) + 100, np.float32); smp[:len(sm)] = sm
    dirs = np.array([[0, 1.0], [0.3827, 0.9239], [0.1951, 0.9808], [0.9239, 0.3827], [0.7071, 0.7071], [0.3827, -0.9239], [0.8315, 0.5556], [0.8315, -0.5556],
                     [0.5556, -0.8315], [0.9808, 0.1951], [0.9239, -0.3827], [0.7071, -0.7071], [0.5556, 0.8315], [0.9808, -0.1951], [1.0, 0.0], [0.1951, -0.9808]], np.float32)
    f32 = np.float32
    ref = np.zeros((h, w), np.float32)
    n2 = n3 = n4 = 0
    dw1 = f32(0.75); dw2 = dw1 * dw1
    ag0, ag1, ag2 = imgs[1], imgs[2], imgs[3]
    for y4 in range(0, h, 4 * pot):
        for x4 in range(0, w, 4 * pot):
            d4 = dirs[rp[n2] & 15]; c4 = (f32(0), -1); any_below = False
            for y3 in range(y4, min(y4 + 4 * pot, h), 2 * pot):
                for x3 in range(x4, min(x4 + 4 * pot, w), 2 * pot):
                    d3 = dirs[rp[n2] & 15]; c3 = (f32(0), -1); any2 = False
                    for y2 in range(y3, min(y3 + 2 * pot, h), pot):
                        for x2 in range(x3, min(x3 + 2 * pot, w), pot):
                            d2 = dirs[rp[n2] & 15]; c2 = (f32(0), -1)
                            for yf in range(y2, min(y2 + pot, h)):
                                for xf in range(x2, min(x2 + pot, w)):
                                    if xf < 4 or xf >= w - 5 or yf < 4 or yf > h - 4:
                                        continue
                                    idx = xf + w * yf
                                    th0 = smp[(xf >> 5) + (yf >> 5) * (w // 32)]; th1 = th0 * dw1; th2 = th1 * dw2
                                    gx, gy = dI[idx, 1], dI[idx, 2]
                                    if ag0[idx] > th0 * thf:
                                        dn = abs(f32(gx * d2[0]) + f32(gy * d2[1]))
                                        if dn > c2[0]: c2 = (dn, idx)
                                    if ag1[int(f32(xf) * f32(0.5) + f32(0.25)) + int(f32(yf) * f32(0.5) + f32(0.25)) * (w // 2)] > th1 * thf:
                                        dn = abs(f32(gx * d3[0]) + f32(gy * d3[1]))
                                        if dn > c3[0]: c3 = (dn, idx)
                                    if ag2[int(float(f32(xf) * f32(0.25)) + 0.125) + int(float(f32(yf) * f32(0.25)) + 0.125) * (w // 4)] > th2 * thf:
                                        dn = abs(f32(gx * d4[0]) + f32(gy * d4[1]))
                                        if dn > c4[0]: c4 = (dn, idx)
                            if c2[1] > 0:
                                ref.flat[c2[1]] = 1; n2 += 1; any2 = True
                    if not any2 and c3[1] > 0:
                        ref.flat[c3[1]] = 2; n3 += 1; any_below = True
                    any_below |= any2
            if not any_below and c4[1] > 0:
                ref.flat[c4[1]] = 4; n4 += 1
    assert np.array_equal(got, ref) and list(n) == [n2, n3, n4] and n2 > 300 and n3 > 0
    # the stripes really make "above threshold but nothing selected" cells (what forces the GPU's count -> scan -> select scheme to iterate)
    th = np.repeat(np.repeat(sm.reshape(h // 32, w // 32), 32, 0), 32, 1)
    a0 = ag0.reshape(h, w)
    assert sum((a0[y:y + 3, x:x + 3] > th[y:y + 3, x:x + 3]).any() and not (got[y:y + 3, x:x + 3] == 1).any() for y in range(36, 75, 3) for x in range(48, 93, 3)) > 0
    # makeMaps: sub-selection drops the rank-th selected pixel when randomPattern[rank] > 255 * quotia
    m, num, newpot = orc.pixsel_make_maps(*imgs, w, h, rp, 200.0, 3, 0, 1.0)
    tot = int(n.sum()); q = np.float32(200.0) / np.float32(tot)
    nz = np.flatnonzero(got)
    keep = rp[:len(nz)] <= np.uint8(int(np.float32(255) * q))
    assert num == keep.sum() and np.array_equal(np.flatnonzero(m), nz[keep]) and newpot == max(int(np.sqrt(np.float32(tot * 16) / np.float32(200.0)) - 1), 1)
    # FusedWithMask
    mask = np.zeros((h, w), np.float32); mask[h // 3:, w // 5:] = np.random.RandomState(5).randint(0, 200, size=(h - h // 3, w - w // 5))
    fm, fn, qm = orc.pixsel_fuse_mask(mask, draws, got)
    assert fn[0] == (fm == 1).sum() and fn[1] == (fm == 2).sum() and fn[2] == 0
    vals = np.sort(mask[mask != 0].astype(int))
    assert abs(qm[0] - vals[len(vals) // 2]) <= 1 and qm[1] == vals.max()
    rs = (draws % 1000).astype(np.float32).reshape(h, w) / np.float32(1000.0)
    assert np.array_equal(fm == 1, ((got == 1) & ~((rs > 0.5) & (mask < qm[0] // 3))) | ((got == 2) & (rs.astype(np.float64) < 0.6) & (mask > qm[0] + (qm[1] - qm[0]) // 2))
                          | ((got != 1) & (got != 2) & (rs.astype(np.float64) < 0.01) & (mask > qm[0])))


def test_marginalize_frame_matches_dense_schur_and_solution_identity(small_window):
    """EnergyFunctional::marginalizeFrame restatement vs an independent numpy formulation: (i) the permuted, prior-augmented, scaled Schur complement
    written with numpy's inverse; (ii) the defining property — solving the marginalised system gives the same remaining variables as solving the full
    one (frame prior included) and dropping the frame's 8 entries."""
    win = small_window
    st6 = synth.perturbed_poses(win, sigma_t=0.003, sigma_r=0.0003)
    aff = [(0.0, 0.0), (0.01, 1.0), (-0.02, -2.0), (0.015, 0.5)]
    ba = orc.ba_from_window(win, "f32", state6=st6, aff=aff)
    ba.linearize_all(False); ba.apply_res()
    ba.marginalize_points((np.arange(len(win.host)) % 3 == 0).astype(np.uint8))
    HM, bM = ba.get_prior()
    n = ba.n
    rng = np.random.RandomState(0)
    A = rng.randn(n, n)
    HM = HM + 1e-3 * np.abs(HM).max() * (A @ A.T) / n                # strictly positive definite, still dominated by the real prior
    ba.set_prior(HM, bM)
    for idx in range(win.W):
        H_o, b_o = ba.marginalize_frame(idx)
        fr = ba.frame(idx)
        prior = np.zeros(8)
        if idx == 0:
            prior[:3], prior[3:6], prior[6:] = 1e10, 1e11, 1e14       # frameID 0: setting_initial{Trans,Rot,Aff}Prior (settings.cpp:58-61)
        else:
            prior[6], prior[7] = 1e12, 1e8                            # setting_affineOptModeA / B (settings.cpp:128-129)
        keep = [i for i in range(n) if i < 4 or (i - 4) // 8 != idx]
        fr_i = [4 + 8 * idx + k for k in range(8)]
        Hf = HM.copy(); bf = bM.copy()
        Hf[np.ix_(fr_i, fr_i)] += np.diag(prior)
        bf[fr_i] += prior * fr["state"][:8]                           # delta_prior = state (getPriorZero() = 0)
        p = keep + fr_i
        Hp, bp = Hf[np.ix_(p, p)], bf[p]
        S = np.sqrt(np.abs(np.diag(Hp)) + 10)
        Hs, bs = Hp / S[:, None] / S[None, :], bp / S
        nd = n - 8
        D_inv = np.linalg.inv(Hs[nd:, nd:])
        Hm = Hs[:nd, :nd] - Hs[nd:, :nd].T @ D_inv @ Hs[nd:, :nd]
        bm = bs[:nd] - Hs[nd:, :nd].T @ D_inv @ bs[nd:]
        Hm = Hm * S[:nd, None] * S[None, :nd]
        bm = bm * S[:nd]
        Hm = 0.5 * (Hm + Hm.T)
        assert rel_err(H_o, Hm) < 1e-9 and rel_err(b_o, bm) < 1e-9
        assert np.abs(H_o - H_o.T).max() == 0
        x_full = np.linalg.solve(Hp, bp)
        x_marg = np.linalg.solve(H_o, b_o)
        assert np.abs(x_marg - x_full[:nd]).max() < 1e-6 * np.abs(x_full[:nd]).max()


def test_l_and_m_energy_closed_forms_and_energy_test_branch():
    """a13 on the oracle: calcLEnergyF_MT / calcMEnergyF (EnergyFunctional.cpp:320-415) against their closed forms in numpy, and the accept / reject branch
    of FullSystem::optimize (setting_forceAceptStep = false, FullSystemOptimize.cpp:511-541): once solveSystemF's fixed lambda (SOLVER_FIX_LAMBDA,
    EnergyFunctional.cpp:779) reproduces a rejected step, every later iteration is rejected too — the run ends at the last accepted state."""
    win = synth.make_window(w=320, h=240, W=4, P=400, seed=7)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    rng = np.random.RandomState(2)
    n = 8 * win.W + 4
    A = rng.randn(n, n); HM = A @ A.T * 1e3; bM = rng.randn(n) * 1e2
    has_prior = (rng.rand(len(win.host)) < 0.3).astype(np.int32)
    idz = (win.idepth * (1 + 1e-3 * rng.randn(len(win.host)))).astype(np.float32)
    orc.lib().orc_set_sum_mode(0)
    ba = orc.BA(win.W, len(win.host), win.w, win.h, win.K)
    for i in range(win.W):
        dI, _ = orc.make_images(win.images[i], 1)
        ba.set_frame(i, dI, win.world_to_cam[i], state6=st6[i])
    ba.set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights, has_prior=has_prior, idepth_zero=idz)
    K = np.asarray(win.K, np.float64)
    ba.set_calib_zero(K * (1 + np.array([1e-4, -1e-4, 2e-4, -2e-4])))
    ba.set_residuals(win.exists)
    ba.set_prior(HM, bM)
    ba.prepare()
    # M energy: delta . (2 bM + HM delta), delta = [cDeltaF | state - state_zero per frame]
    cz = K * (1 + np.array([1e-4, -1e-4, 2e-4, -2e-4]))
    cd = np.float32(np.array([K[0] / 50, K[1] / 50, K[2] / 50, K[3] / 50]) - np.array([np.float32(1 / 50.) * cz[0], np.float32(1 / 50.) * cz[1], np.float32(1 / 50.) * cz[2], np.float32(1 / 50.) * cz[3]]))
    delta = np.r_[cd.astype(np.float64), np.concatenate([ba.frame(i)["state"][:8] - ba.frame(i)["state_zero"][:8] for i in range(win.W)])]
    M_np = delta @ (2 * bM + HM @ delta)
    assert abs(ba.calc_m_energy() - M_np) < 1e-4 * abs(M_np) + 1e-9          # cDeltaF is a float difference of nearly equal numbers: ~1e-5 of its value
    # L energy: frame priors (frame 0: 1e10 / 1e11 / 1e14, others: a, b priors 1e12 / 1e8) on delta_prior = state, the calibration prior 5e9 on cDeltaF,
    # and deltaF^2 * 2500 for the points with a depth prior
    L_np = 0.0
    for i in range(win.W):
        s = ba.frame(i)["state"][:8]
        p = np.r_[[1e10] * 3, [1e11] * 3, 1e14, 1e14] if i == 0 else np.r_[[0.0] * 6, 1e12, 1e8]
        L_np += float((s * p * s).sum())
    L_np += float((cd.astype(np.float64) ** 2 * 5e9).sum())
    d = (win.idepth.astype(np.float32) - idz).astype(np.float64)
    L_np += float((d * d * 2500.0 * (has_prior != 0)).sum())
    assert abs(ba.calc_l_energy() - L_np) < 1e-4 * abs(L_np)
    # the energy test
    ba.set_settings(force_accept_step=False)
    ba.optimize(6)
    assert 0 <= ba.n_rejected() <= 6
