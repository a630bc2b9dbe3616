"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY — parity unpinned, see orc_common.h).
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int)
c_u8p = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


def _p(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(t)


def dp(a):
    return _p(a, c_dp)


def fp(a):
    return _p(a, c_fp)


def ip(a):
    return _p(a, c_ip)


def u8p(a):
    return _p(a, c_u8p)


def lib(kind="f32"):
    """kind: 'f32' (reference-faithful pointwise fp32), 'f64' (all-fp64 ground truth), 'fast' (timed baseline)."""
    if kind in _LIBS:
        return _LIBS[kind]
    path = os.path.join(_DIR, "liboracle_%s.so" % kind)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.orc_trk_create.restype = C.c_void_p
    L.orc_trk_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_trk_destroy.argtypes = [C.c_void_p]
    L.orc_trk_set_ref.argtypes = [C.c_void_p, c_fp, C.c_int, c_fp, c_fp, c_fp, c_fp]
    L.orc_trk_set_pc.argtypes = [C.c_void_p, c_fp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]
    L.orc_trk_get_pc.argtypes = [C.c_void_p, C.c_int, c_fp, c_fp, c_fp, c_fp]
    L.orc_trk_get_pc.restype = C.c_int
    L.orc_trk_get_depth.argtypes = [C.c_void_p, C.c_int, c_fp, c_fp]
    L.orc_trk_calc_res.argtypes = [C.c_void_p, c_fp, C.c_int, c_dp, c_dp, c_fp, C.c_float, c_dp]
    L.orc_trk_calc_gs.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, c_dp, c_dp]
    L.orc_trk_track.argtypes = [C.c_void_p, c_fp, c_dp, c_dp, c_dp, c_fp, C.c_int, c_dp, c_dp, c_dp]
    L.orc_trk_track.restype = C.c_int
    L.orc_trk_counter.argtypes = [C.c_void_p, C.c_int]
    L.orc_trk_counter.restype = C.c_long
    L.orc_make_images.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp]
    L.orc_pyr_offset.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_pyr_offset.restype = C.c_long
    L.orc_pyr_levels.argtypes = [C.c_int, C.c_int]
    L.orc_pyr_levels.restype = C.c_int
    L.orc_se3_exp.argtypes = [c_dp, c_dp]
    L.orc_se3_log.argtypes = [c_dp, c_dp]
    L.orc_se3_mul.argtypes = [c_dp, c_dp, c_dp]
    L.orc_se3_inv.argtypes = [c_dp, c_dp]
    L.orc_se3_adj.argtypes = [c_dp, c_dp]
    L.orc_ldlt_solve.argtypes = [C.c_int, c_dp, c_dp, c_dp]
    L.orc_set_sum_mode.argtypes = [C.c_int]
    L.orc_ba_create.restype = C.c_void_p
    L.orc_ba_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
    L.orc_ba_destroy.argtypes = [C.c_void_p]
    L.orc_ba_set_frame.argtypes = [C.c_void_p, C.c_int, c_fp, c_dp, C.c_double, C.c_double, C.c_float, C.c_float, C.c_int, c_dp]
    L.orc_ba_set_points.argtypes = [C.c_void_p, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip]
    L.orc_ba_set_residuals.argtypes = [C.c_void_p, c_u8p]
    L.orc_ba_set_adjoints.argtypes = [C.c_void_p]
    L.orc_ba_set_precalc.argtypes = [C.c_void_p]
    L.orc_ba_linearize_all.argtypes = [C.c_void_p, C.c_int]
    L.orc_ba_linearize_all.restype = C.c_double
    L.orc_ba_apply_res.argtypes = [C.c_void_p]
    L.orc_ba_accumulate.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp]
    L.orc_ba_accumulate_sc.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp]
    L.orc_ba_solve_system.argtypes = [C.c_void_p, C.c_int, C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.orc_ba_do_step.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_ba_do_step.restype = C.c_int
    L.orc_ba_optimize.argtypes = [C.c_void_p, C.c_int]
    L.orc_ba_optimize.restype = C.c_double
    L.orc_ba_marginalize_points.argtypes = [C.c_void_p, c_u8p, c_dp, c_dp, c_dp, c_dp]
    L.orc_ba_get_residual.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_ip, c_dp, c_fp]
    L.orc_ba_get_slots.argtypes = [C.c_void_p, C.POINTER(C.c_int8), c_u8p, c_fp, c_fp]
    L.orc_ba_get_points.argtypes = [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
    L.orc_ba_get_frame.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_fp]
    L.orc_ba_get_calib.argtypes = [C.c_void_p, c_dp]
    L.orc_ba_get_precalc.argtypes = [C.c_void_p, c_fp]
    L.orc_ba_get_adjoints.argtypes = [C.c_void_p, c_dp, c_dp, c_fp]
    L.orc_ba_get_prior.argtypes = [C.c_void_p, c_dp, c_dp]
    L.orc_ba_set_prior.argtypes = [C.c_void_p, c_dp, c_dp]
    L.orc_ba_set_options.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_ba_get_timers.argtypes = [C.c_void_p, c_dp]
    L.orc_ba_counts.argtypes = [C.c_void_p, C.c_int]
    L.orc_ba_counts.restype = C.c_int
    L.orc_ba_set_idepth.argtypes = [C.c_void_p, c_fp]
    L.orc_ba_get_center_projected.argtypes = [C.c_void_p, c_fp]
    L.orc_ba_set_linearize_mt.argtypes = [C.c_void_p, C.c_int]
    L.orc_ba_get_frame_state_zero.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.orc_ba_set_frame_full.argtypes = [C.c_void_p, C.c_int, c_fp, c_dp, c_dp, c_dp, C.c_float, C.c_float, C.c_int]
    L.orc_ba_set_idepth_zero.argtypes = [C.c_void_p, c_fp]
    L.orc_ba_set_calib_zero.argtypes = [C.c_void_p, c_dp]
    L.orc_ba_marginalize_frame.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp]
    L.orc_ba_get_precalc_rt.argtypes = [C.c_void_p, c_fp, c_fp]
    L.orc_init_calc_res_and_gs.argtypes = [c_fp, c_fp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_double, C.c_float, C.c_float, C.c_float, C.c_int,
                                           c_fp, c_fp, c_fp, c_fp, c_u8p, c_fp, c_fp, c_u8p, c_fp, c_fp, c_fp, c_fp, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.orc_init_do_step.argtypes = [C.c_int, c_u8p, c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp]
    L.orc_pixsel_make_hists.argtypes = [c_fp, C.c_int, C.c_int, c_fp, c_fp]
    L.orc_dist_make_map.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
    L.orc_imm_create.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp]
    L.orc_imm_trace.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip, c_fp, c_fp, c_fp]
    L.orc_imm_optimize.argtypes = [C.c_int, C.POINTER(c_fp), C.c_int, C.c_int, c_fp, c_fp, c_fp, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int, c_ip, c_fp, c_u8p]
    L.orc_dense_bbox.argtypes = [c_fp, C.c_int, C.c_int, C.c_float, c_ip]
    L.orc_dense_make_map.argtypes = [c_fp, c_fp, c_u8p, C.c_int, C.c_int, c_fp, C.c_float, c_ip, C.c_float, C.c_float,
                                     C.c_float, C.c_float, c_dp, c_ip, c_ip, c_fp, c_fp, c_u8p, c_ip]
    L.orc_dense_make_map.restype = C.c_int
    _LIBS[kind] = L
    return L


# ------------------------------------------------------------------ convenience wrappers
def se3_exp(xi, kind="f32"):
    T = np.zeros(12)
    lib(kind).orc_se3_exp(dp(np.ascontiguousarray(xi, np.float64)), dp(T))
    return T.reshape(3, 4)


def se3_log(T, kind="f32"):
    xi = np.zeros(6)
    lib(kind).orc_se3_log(dp(np.ascontiguousarray(T, np.float64).reshape(-1)), dp(xi))
    return xi


def se3_mul(A, B, kind="f32"):
    Cc = np.zeros(12)
    lib(kind).orc_se3_mul(dp(np.ascontiguousarray(A, np.float64).reshape(-1)), dp(np.ascontiguousarray(B, np.float64).reshape(-1)), dp(Cc))
    return Cc.reshape(3, 4)


def se3_inv(A, kind="f32"):
    Cc = np.zeros(12)
    lib(kind).orc_se3_inv(dp(np.ascontiguousarray(A, np.float64).reshape(-1)), dp(Cc))
    return Cc.reshape(3, 4)


def se3_adj(A, kind="f32"):
    Ad = np.zeros(36)
    lib(kind).orc_se3_adj(dp(np.ascontiguousarray(A, np.float64).reshape(-1)), dp(Ad))
    return Ad.reshape(6, 6)


def make_images(img, levels, kind="f32"):
    """returns (dI_all [npix,3] float32, abs_all [npix]) with pyramid levels concatenated."""
    h, w = img.shape
    L = lib(kind)
    tot = L.orc_pyr_offset(w, h, levels)
    dI = np.zeros((tot, 3), np.float32)
    ab = np.zeros(tot, np.float32)
    L.orc_make_images(fp(np.ascontiguousarray(img, np.float32)), w, h, levels, None, fp(dI), fp(ab))
    return dI, ab


# ------------------------------------------------------------------ immature points
def pixsel_make_hists(absg0, w, h, kind="f32"):
    nb = (w // 32) * (h // 32)
    ths, sm = np.zeros(nb, np.float32), np.zeros(nb, np.float32)
    lib(kind).orc_pixsel_make_hists(fp(np.ascontiguousarray(absg0, np.float32)), w, h, fp(ths), fp(sm))
    return ths, sm


def pixsel_libc_tables(n, kind="f32"):
    """randomPattern[n] (PixelSelector ctor) and the first n rand() draws after srand(3141592) (FusedWithMask), from this machine's libc"""
    rp, dr = np.zeros(n, np.uint8), np.zeros(n, np.int32)
    lib(kind).orc_pixsel_libc_tables(n, rp.ctypes.data_as(C.POINTER(C.c_ubyte)), ip(dr))
    return rp, dr


def _pixsel_imgs(dI, ag0, ag1, ag2):
    f = lambda a: np.ascontiguousarray(a, np.float32)
    return f(dI), f(ag0), f(ag1), f(ag2)


def pixsel_select(dI, ag0, ag1, ag2, w, h, thsSmoothed, randomPattern, pot, thFactor=1.0, kind="f32"):
    dI, ag0, ag1, ag2 = _pixsel_imgs(dI, ag0, ag1, ag2)
    m, n = np.zeros((h, w), np.float32), np.zeros(3, np.int32)
    rp = np.ascontiguousarray(randomPattern, np.uint8)
    sm = np.zeros((w // 32) * (h // 32) + 100, np.float32)      # the reference's allocation; the tail it never fills is defined as 0 (orc_pixsel.c)
    sm[:len(thsSmoothed)] = thsSmoothed
    lib(kind).orc_pixsel_select(fp(dI), fp(ag0), fp(ag1), fp(ag2), w, h, fp(sm), rp.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                pot, C.c_float(thFactor), fp(m), ip(n))
    return m, n


def pixsel_make_maps(dI, ag0, ag1, ag2, w, h, randomPattern, density, potential, recursionsLeft=1, thFactor=1.0, kind="f32"):
    dI, ag0, ag1, ag2 = _pixsel_imgs(dI, ag0, ag1, ag2)
    m, pot = np.zeros((h, w), np.float32), np.array([potential], np.int32)
    rp = np.ascontiguousarray(randomPattern, np.uint8)
    L = lib(kind)
    L.orc_pixsel_make_maps.restype = C.c_int
    num = L.orc_pixsel_make_maps(fp(dI), fp(ag0), fp(ag1), fp(ag2), w, h, rp.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_float(density), recursionsLeft, C.c_float(thFactor), ip(pot), fp(m))
    return m, int(num), int(pot[0])


def pixsel_fuse_mask(mask, draws, m, kind="f32"):
    m = np.ascontiguousarray(m, np.float32).copy()
    n, qm = np.zeros(3, np.int32), np.zeros(2, np.int32)
    lib(kind).orc_pixsel_fuse_mask(fp(np.ascontiguousarray(mask, np.float32)), ip(np.ascontiguousarray(draws, np.int32)), m.size, fp(m), ip(n), ip(qm))
    return m, n, qm


def pixsel_make_maps_lidar(dI, ag0, ag1, ag2, w, h, randomPattern, mask, draws, potential, thFactor=1.0, kind="f32"):
    dI, ag0, ag1, ag2 = _pixsel_imgs(dI, ag0, ag1, ag2)
    m = np.zeros((h, w), np.float32)
    rp = np.ascontiguousarray(randomPattern, np.uint8)
    L = lib(kind)
    L.orc_pixsel_make_maps_lidar.restype = C.c_int
    num = L.orc_pixsel_make_maps_lidar(fp(dI), fp(ag0), fp(ag1), fp(ag2), w, h, rp.ctypes.data_as(C.POINTER(C.c_ubyte)), fp(np.ascontiguousarray(mask, np.float32)),
                                       ip(np.ascontiguousarray(draws, np.int32)), C.c_float(thFactor), potential, fp(m))
    return m, int(num)


def dist_make_map(w1, h1, frame, host, u, v, idepth, KRKi, Kt, kind="f32"):
    """CoarseDistanceMap::makeDistanceMap -> [h1, w1] float map (1000 = farther than 39)"""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    out = np.zeros((h1, w1), np.float32)
    a = [f(x) for x in (u, v, idepth, KRKi, Kt)]
    lib(kind).orc_dist_make_map(w1, h1, frame, len(a[0]), ip(np.ascontiguousarray(host, np.int32)), *[fp(x) for x in a], fp(out))
    return out


def imm_create(dI_host, w, h, u, v, kind="f32"):
    """ImmaturePoint ctor for n integer pixel positions. dI_host: [w*h,3] level-0 texels. -> color[n,8], weights[n,8], gradH[n,3], energyTH[n]"""
    n = len(u)
    color, weights, gradH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    ui, vi = np.ascontiguousarray(u, np.int32), np.ascontiguousarray(v, np.int32)
    lib(kind).orc_imm_create(fp(np.ascontiguousarray(dI_host, np.float32)), w, h, n, ip(ui), ip(vi), fp(color), fp(weights), fp(gradH), fp(eth))
    return color, weights, gradH, eth


def imm_trace(dI_new, w, h, u, v, color, weights, gradH, energyTH, host_idx, KRKi, Kt, aff, idmin, idmax, status, quality, kind="f32"):
    """traceOn for every point (in/out arrays are copied). -> idmin, idmax, status, quality, lastUV[n,2], lastInterval[n]"""
    n = len(u)
    f = lambda a: np.ascontiguousarray(a, np.float32).copy()
    idmin, idmax, quality = f(idmin), f(idmax), f(quality)
    status = np.ascontiguousarray(status, np.int32).copy()
    uv, li = np.zeros((n, 2), np.float32), np.zeros(n, np.float32)
    hi = np.ascontiguousarray(host_idx, np.int32)
    a = [f(x) for x in (u, v, color, weights, gradH, energyTH)]
    k = [f(x) for x in (KRKi, Kt, aff)]
    lib(kind).orc_imm_trace(fp(np.ascontiguousarray(dI_new, np.float32)), w, h, n, *[fp(x) for x in a], ip(hi), *[fp(x) for x in k],
                            fp(idmin), fp(idmax), ip(status), fp(quality), fp(uv), fp(li))
    return idmin, idmax, status, quality, uv, li


def imm_optimize(dI_list, w, h, K, Rt, aff, host, u, v, color, weights, energyTH, idmin, idmax, min_obs, kind="f32"):
    """optimizeImmaturePoint for every point. dI_list: level-0 texels of the W window frames. -> result[n], idepth[n], res_in[n,W]"""
    W, n = len(dI_list), len(u)
    f = lambda a: np.ascontiguousarray(a, np.float32)
    imgs = [f(d) for d in dI_list]
    arr = (c_fp * W)(*[fp(d) for d in imgs])
    res, idp, rin = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, W), np.uint8)
    a = [f(x) for x in (u, v, color, weights, energyTH, idmin, idmax)]
    Kf, Rtf, aff_f, hi = f(K), f(Rt), f(aff), np.ascontiguousarray(host, np.int32)
    lib(kind).orc_imm_optimize(W, arr, w, h, fp(Kf), fp(Rtf), fp(aff_f), n, ip(hi), *[fp(x) for x in a], int(min_obs), ip(res), fp(idp), u8p(rin))
    return res, idp, rin


# ------------------------------------------------------------------ two-frame initialiser
def init_calc_res_and_gs(colorRef, colorNew, wl, hl, K4, refToNew, aff, pts, alphaW=150.0 * 150.0, alphaK=2.5 * 2.5, couplingWeight=1.0, kind="f32"):
    """CoarseInitializer::calcResAndGS on one level. pts: dict of arrays u, v, idepth_new, iR, isGood, energy[n,2], outlierTH, lastHessian_new, Jb[n,10]
    (the last two are in/out). -> dict(H, b, Hsc, bsc, E3, isGood_new, energy_new, maxstep, lastHessian_new, Jb)"""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    T = np.asarray(refToNew, np.float64).reshape(3, 4)
    fx, fy, cx, cy = [float(x) for x in K4]
    Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1.0]])
    # (R * K^-1) in double, term by term in the order of the C++ host code, then cast (a BLAS matmul may fuse/reorder the three terms)
    RKi = f([float(T[i, 0]) * float(Ki[0, j]) + float(T[i, 1]) * float(Ki[1, j]) + float(T[i, 2]) * float(Ki[2, j]) for i in range(3) for j in range(3)])
    t = f(T[:, 3])
    aff2 = f([np.exp(aff[0]), aff[1]])
    tlog = f(se3_log(T, kind)[:3])
    n = len(pts["u"])
    ig = np.ascontiguousarray(pts["isGood"], np.uint8)
    ign, en, ms = np.zeros(n, np.uint8), np.zeros((n, 2), np.float32), np.zeros(n, np.float32)
    lh, jb = f(pts["lastHessian_new"]).copy(), f(pts["Jb"]).copy()
    H, b, Hs, bs, E3 = np.zeros(64), np.zeros(8), np.zeros(64), np.zeros(8), np.zeros(3)
    a = [f(pts[k]) for k in ("u", "v", "idepth_new", "iR")]
    eng, oth = f(pts["energy"]), f(pts["outlierTH"])
    lib(kind).orc_init_calc_res_and_gs(fp(f(colorRef)), fp(f(colorNew)), wl, hl, fp(f(K4)), fp(RKi), fp(t), fp(aff2), fp(tlog), float((T[:, 3] ** 2).sum()),
                                       alphaW, alphaK, couplingWeight, n, *[fp(x) for x in a], u8p(ig), fp(eng), fp(oth), u8p(ign), fp(en), fp(ms), fp(lh), fp(jb),
                                       dp(H), dp(b), dp(Hs), dp(bs), dp(E3))
    return dict(H=H.reshape(8, 8), b=b, Hsc=Hs.reshape(8, 8), bsc=bs, E3=E3, isGood_new=ign, energy_new=en, maxstep=ms, lastHessian_new=lh, Jb=jb)


def init_do_step(isGood, Jb, maxstep, idepth, lam, inc, idepth_new, kind="f32"):
    f = lambda a: np.ascontiguousarray(a, np.float32)
    out = f(idepth_new).copy()
    lib(kind).orc_init_do_step(len(out), u8p(np.ascontiguousarray(isGood, np.uint8)), fp(f(Jb)), fp(f(maxstep)), fp(f(idepth)), float(lam), fp(f(inc)), fp(out))
    return out


class Tracker:
    def append_plane_points(self, mask, dirv, dis, ref_color, rect):
        self.L.orc_trk_append_plane_points.argtypes = [C.c_void_p, c_fp, c_fp, C.c_float, C.c_int, c_ip]
        self.L.orc_trk_append_plane_points.restype = C.c_int
        return self.L.orc_trk_append_plane_points(self.h_, fp(np.ascontiguousarray(mask, np.float32)), fp(np.ascontiguousarray(dirv, np.float32)), float(dis), int(ref_color),
                                                  ip(np.ascontiguousarray(rect, np.int32)))

    def set_affine_modes(self, a, b):
        """setting_affineOptModeA / B (< 0: fixed)"""
        self.L.orc_trk_set_affine_modes.argtypes = [C.c_void_p, C.c_double, C.c_double]
        self.L.orc_trk_set_affine_modes(self.h_, float(a), float(b))

    def __init__(self, w, h, levels, K, kind="f32"):
        self.L = lib(kind)
        self.w, self.h, self.levels = w, h, levels
        self.h_ = self.L.orc_trk_create(w, h, levels, K[0], K[1], K[2], K[3])
        self._keep = []

    def __del__(self):
        try:
            self.L.orc_trk_destroy(self.h_)
        except Exception:
            pass

    def set_ref(self, dI_ref, Ku, Kv, new_idepth, HdiF):
        self._keep = [np.ascontiguousarray(dI_ref, np.float32)]
        a = [np.ascontiguousarray(x, np.float32) for x in (Ku, Kv, new_idepth, HdiF)]
        self.L.orc_trk_set_ref(self.h_, fp(self._keep[0]), len(a[0]), fp(a[0]), fp(a[1]), fp(a[2]), fp(a[3]))

    def set_pc(self, dI_ref, lvl, u, v, idepth, color):
        self._keep = [np.ascontiguousarray(dI_ref, np.float32)]
        a = [np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, color)]
        self.L.orc_trk_set_pc(self.h_, fp(self._keep[0]), lvl, len(a[0]), fp(a[0]), fp(a[1]), fp(a[2]), fp(a[3]))

    def get_pc(self, lvl):
        n = self.L.orc_trk_get_pc(self.h_, lvl, None, None, None, None)
        out = [np.zeros(n, np.float32) for _ in range(4)]
        self.L.orc_trk_get_pc(self.h_, lvl, *[fp(o) for o in out])
        return out

    def get_depth(self, lvl):
        n = (self.w >> lvl) * (self.h >> lvl)
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self.L.orc_trk_get_depth(self.h_, lvl, fp(a), fp(b))
        return a, b

    def calc_res(self, dI_new, lvl, T, affLL, cutoff):
        T = np.ascontiguousarray(T, np.float64).reshape(3, 4)
        R = np.ascontiguousarray(T[:, :3]).reshape(-1)
        t = np.ascontiguousarray(T[:, 3])
        out = np.zeros(6)
        self.L.orc_trk_calc_res(self.h_, fp(dI_new), lvl, dp(R), dp(t), fp(np.asarray(affLL, np.float32)), cutoff, dp(out))
        return out

    def calc_gs(self, lvl, aff_a, b0):
        H, b = np.zeros(64), np.zeros(8)
        self.L.orc_trk_calc_gs(self.h_, lvl, aff_a, b0, dp(H), dp(b))
        return H.reshape(8, 8), b

    def track(self, dI_new, T0, aff0, ref_aff, exposures, coarsest, min_res=None):
        T = np.ascontiguousarray(T0, np.float64).reshape(-1).copy()
        aff = np.array(aff0, np.float64)
        mr = np.full(5, np.nan) if min_res is None else np.asarray(min_res, np.float64)
        lr, lf = np.zeros(5), np.zeros(3)
        ok = self.L.orc_trk_track(self.h_, fp(dI_new), dp(T), dp(aff), dp(np.asarray(ref_aff, np.float64)),
                                  fp(np.asarray(exposures, np.float32)), coarsest, dp(mr), dp(lr), dp(lf))
        return ok, T.reshape(3, 4), aff, lr, lf


class BA:
    """Sliding-window BA oracle over a synth.Window-like description."""

    def __init__(self, W, P, w, h, K, kind="f32"):
        self.L = lib(kind)
        self.W, self.P, self.w, self.h = W, P, w, h
        self.n = 8 * W + 4
        self.h_ = self.L.orc_ba_create(W, P, w, h, K[0], K[1], K[2], K[3])
        self._keep = {}

    def __del__(self):
        try:
            self.L.orc_ba_destroy(self.h_)
        except Exception:
            pass

    def set_frame(self, i, dI0, evalPT, aff=(0.0, 0.0), exposure=1.0, th=8 * 8 * 8.0, frame_id=None, state6=None):
        self._keep[i] = np.ascontiguousarray(dI0, np.float32)
        s6 = None if state6 is None else np.ascontiguousarray(state6, np.float64)
        self.L.orc_ba_set_frame(self.h_, i, fp(self._keep[i]), dp(np.ascontiguousarray(evalPT, np.float64).reshape(-1)),
                                aff[0], aff[1], exposure, th, i if frame_id is None else frame_id, dp(s6))

    def set_frame_full(self, i, dI0, evalPT, state, state_zero, exposure=1.0, th=8 * 8 * 8.0, frame_id=None):
        self._keep[i] = np.ascontiguousarray(dI0, np.float32)
        self.L.orc_ba_set_frame_full(self.h_, i, fp(self._keep[i]), dp(np.ascontiguousarray(evalPT, np.float64).reshape(-1)),
                                     dp(np.ascontiguousarray(state, np.float64)), dp(np.ascontiguousarray(state_zero, np.float64)),
                                     exposure, th, i if frame_id is None else frame_id)

    def set_points(self, host, u, v, idepth, color, weights, has_prior=None, idepth_zero=None):
        a = [np.ascontiguousarray(host, np.int32)] + [np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, color, weights)]
        hp = None if has_prior is None else np.ascontiguousarray(has_prior, np.int32)
        self.L.orc_ba_set_points(self.h_, ip(a[0]), fp(a[1]), fp(a[2]), fp(a[3]), fp(a[4]), fp(a[5]), ip(hp))
        if idepth_zero is not None:
            self.L.orc_ba_set_idepth_zero(self.h_, fp(np.ascontiguousarray(idepth_zero, np.float32)))

    def set_calib_zero(self, calib_zero_scaled):
        self.L.orc_ba_set_calib_zero(self.h_, dp(np.ascontiguousarray(calib_zero_scaled, np.float64)))

    def set_prior(self, HM, bM):
        self.L.orc_ba_set_prior(self.h_, dp(np.ascontiguousarray(HM, np.float64)), dp(np.ascontiguousarray(bM, np.float64)))

    def get_prior(self):
        n = self.n
        H, b = np.zeros(n * n), np.zeros(n)
        self.L.orc_ba_get_prior(self.h_, dp(H), dp(b))
        return H.reshape(n, n), b

    def marginalize_frame(self, idx):
        """EnergyFunctional::marginalizeFrame -> (HM, bM) of the window without frame idx"""
        n = self.n - 8
        H, b = np.zeros(n * n), np.zeros(n)
        self.L.orc_ba_marginalize_frame(self.h_, int(idx), dp(H), dp(b))
        return H.reshape(n, n), b

    def set_residuals(self, exists):
        self.L.orc_ba_set_residuals(self.h_, u8p(np.ascontiguousarray(exists, np.uint8)))

    def prepare(self):
        self.L.orc_ba_set_adjoints(self.h_)
        self.L.orc_ba_set_precalc(self.h_)

    def linearize_all(self, fix=False):
        return self.L.orc_ba_linearize_all(self.h_, int(fix))

    def apply_res(self):
        self.L.orc_ba_apply_res(self.h_)

    def accumulate(self, mode, want13=False):
        H, b = np.zeros(self.n * self.n), np.zeros(self.n)
        h13 = np.zeros(self.W * self.W * 169) if want13 else None
        self.L.orc_ba_accumulate(self.h_, mode, dp(H), dp(b), dp(h13))
        H = H.reshape(self.n, self.n)
        return (H, b, h13.reshape(self.W * self.W, 13, 13)) if want13 else (H, b)

    def accumulate_sc(self, shift=True):
        H, b = np.zeros(self.n * self.n), np.zeros(self.n)
        self.L.orc_ba_accumulate_sc(self.h_, int(shift), dp(H), dp(b))
        return H.reshape(self.n, self.n), b

    def solve_system(self, iteration, lam=1e-5, debug=False):
        x = np.zeros(self.n)
        if debug:
            HA, bA, Hs, bs = np.zeros(self.n ** 2), np.zeros(self.n), np.zeros(self.n ** 2), np.zeros(self.n)
            self.L.orc_ba_solve_system(self.h_, iteration, lam, dp(x), dp(HA), dp(bA), dp(Hs), dp(bs))
            return x, HA.reshape(self.n, self.n), bA, Hs.reshape(self.n, self.n), bs
        self.L.orc_ba_solve_system(self.h_, iteration, lam, dp(x), None, None, None, None)
        return x

    def do_step(self, f=1.0):
        return self.L.orc_ba_do_step(self.h_, f, f, f, f, f)

    def optimize(self, its=6):
        return self.L.orc_ba_optimize(self.h_, its)

    def marginalize_points(self, flags):
        n = self.n
        M, Mb, Ms, Mbs = np.zeros(n * n), np.zeros(n), np.zeros(n * n), np.zeros(n)
        self.L.orc_ba_marginalize_points(self.h_, u8p(np.ascontiguousarray(flags, np.uint8)), dp(M), dp(Mb), dp(Ms), dp(Mbs))
        return M.reshape(n, n), Mb, Ms.reshape(n, n), Mbs

    def residual(self, p, t):
        J, jp, rtz = np.zeros(74), np.zeros(8), np.zeros(8)
        st = np.zeros(3, np.int32)
        en = np.zeros(3)
        pr = np.zeros(19, np.float32)
        self.L.orc_ba_get_residual(self.h_, p, t, dp(J), dp(jp), dp(rtz), ip(st), dp(en), fp(pr))
        return dict(J=J, JpJdF=jp, rtz=rtz, state=st, energy=en, proj=pr)

    def slots(self):
        n = self.P * self.W
        st = np.zeros(n, np.int8)
        ac = np.zeros(n, np.uint8)
        jp = np.zeros(n * 8, np.float32)
        en = np.zeros(n, np.float32)
        self.L.orc_ba_get_slots(self.h_, st.ctypes.data_as(C.POINTER(C.c_int8)), u8p(ac), fp(jp), fp(en))
        return st.reshape(self.P, self.W), ac.reshape(self.P, self.W), jp.reshape(self.P, self.W, 8), en.reshape(self.P, self.W)

    def center_projected(self):
        o = np.zeros((self.P, self.W, 3), np.float32)
        self.L.orc_ba_get_center_projected(self.h_, fp(o))
        return o

    def points(self):
        P = self.P
        o = {k: np.zeros(P, np.float32) for k in ("idepth", "step", "HdiF", "bdSumF", "Hdd", "bd")}
        o["Hcd"] = np.zeros(4 * P, np.float32)
        o["maxRelBaseline"] = np.zeros(P, np.float32)
        self.L.orc_ba_get_points(self.h_, fp(o["idepth"]), fp(o["step"]), fp(o["HdiF"]), fp(o["bdSumF"]), fp(o["Hdd"]), fp(o["bd"]),
                                 fp(o["Hcd"]), fp(o["maxRelBaseline"]))
        o["Hcd"] = o["Hcd"].reshape(P, 4)
        return o

    def frame(self, f):
        st, w2c, ev = np.zeros(10), np.zeros(12), np.zeros(12)
        th = C.c_float(0)
        self.L.orc_ba_get_frame(self.h_, f, dp(st), dp(w2c), dp(ev), C.byref(th))
        sz = np.zeros(10)
        self.L.orc_ba_get_frame_state_zero(self.h_, f, dp(sz))
        return dict(state=st, state_zero=sz, worldToCam=w2c.reshape(3, 4), evalPT=ev.reshape(3, 4), frameEnergyTH=th.value)

    def calib(self):
        v = np.zeros(4)
        self.L.orc_ba_get_calib(self.h_, dp(v))
        return v

    def precalc(self):
        o = np.zeros((self.W * self.W, 32), np.float32)
        self.L.orc_ba_get_precalc(self.h_, fp(o))
        return o

    def precalc_rt(self):
        """PRE_RTll | PRE_tTll [W*W,12] and PRE_aff_mode [W*W,2] of the current states (index host*W + target)"""
        rt, af = np.zeros((self.W * self.W, 12), np.float32), np.zeros((self.W * self.W, 2), np.float32)
        self.L.orc_ba_get_precalc_rt(self.h_, fp(rt), fp(af))
        return rt, af

    def adjoints(self):
        n = self.W * self.W
        a, b, d = np.zeros((n, 8, 8)), np.zeros((n, 8, 8)), np.zeros((n, 8), np.float32)
        self.L.orc_ba_get_adjoints(self.h_, dp(a), dp(b), fp(d))
        return a, b, d

    def set_options(self, nthreads=6, never_break=False):
        self.L.orc_ba_set_options(self.h_, nthreads, int(never_break))

    def set_settings(self, force_accept_step=True, affine_opt_mode_a=1e12, affine_opt_mode_b=1e8, min_opt_iterations=1):
        """setting_forceAceptStep / setting_affineOptModeA,B / setting_minOptIterations (util/settings.cpp:71,128-129,74)"""
        self.L.orc_ba_set_settings.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        self.L.orc_ba_set_settings(self.h_, int(force_accept_step), float(affine_opt_mode_a), float(affine_opt_mode_b), int(min_opt_iterations))

    def plane_scale_fix(self, localscale, camToTrackingRef, trackingRef_camToWorld):
        self.L.orc_ba_plane_scale_fix.argtypes = [C.c_void_p, C.c_double, c_dp, c_dp]
        self.L.orc_ba_plane_scale_fix(self.h_, float(localscale), dp(np.ascontiguousarray(camToTrackingRef, np.float64).reshape(-1)), dp(np.ascontiguousarray(trackingRef_camToWorld, np.float64).reshape(-1)))

    def sw_gray_optimize(self):
        self.L.orc_ba_sw_gray_optimize.argtypes = [C.c_void_p, c_ip]; self.L.orc_ba_sw_gray_optimize.restype = C.c_double
        n = np.zeros(1, np.int32)
        cost = self.L.orc_ba_sw_gray_optimize(self.h_, ip(n))
        return cost, int(n[0])

    def idepth_zero(self):
        self.L.orc_ba_get_idepth_zero.argtypes = [C.c_void_p, c_fp]
        o = np.zeros(self.P, np.float32)
        self.L.orc_ba_get_idepth_zero(self.h_, fp(o))
        return o

    def calc_l_energy(self):
        self.L.orc_ba_calc_l_energy.argtypes = [C.c_void_p]; self.L.orc_ba_calc_l_energy.restype = C.c_double
        return self.L.orc_ba_calc_l_energy(self.h_)

    def calc_m_energy(self):
        self.L.orc_ba_calc_m_energy.argtypes = [C.c_void_p]; self.L.orc_ba_calc_m_energy.restype = C.c_double
        return self.L.orc_ba_calc_m_energy(self.h_)

    def n_rejected(self):
        self.L.orc_ba_n_rejected.argtypes = [C.c_void_p]; self.L.orc_ba_n_rejected.restype = C.c_int
        return self.L.orc_ba_n_rejected(self.h_)

    def timers(self):
        t = np.zeros(4)
        self.L.orc_ba_get_timers(self.h_, dp(t))
        return t

    def counts(self):
        return [self.L.orc_ba_counts(self.h_, i) for i in range(3)]


def ba_from_window(win, kind="f32", state6=None, aff=None, th=None):
    """Builds a BA oracle from a synth.Window (keyframes 0..W-1)."""
    ba = BA(win.W, len(win.host), win.w, win.h, win.K, kind)
    for i in range(win.W):
        dI, _ = make_images(win.images[i], 1, kind)
        ba.set_frame(i, dI, win.world_to_cam[i], aff=(0.0, 0.0) if aff is None else tuple(aff[i]),
                     th=8 * 8 * 8.0 if th is None else th[i], state6=None if state6 is None else state6[i])
    ba.set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
    ba.set_residuals(win.exists)
    ba.prepare()
    return ba


# ------------------------------------------------------------------ the rest of the two-frame initialiser (orc_initfull.c)
def _knn(fn, u, v, qu, qv, k):
    f = lambda a: np.ascontiguousarray(a, np.float32)
    u, v, qu, qv = f(u), f(v), f(qu), f(qv)
    idx, dist = np.zeros((len(qu), k), np.int32), np.zeros((len(qu), k), np.float32)
    fn.argtypes = [C.c_int, c_fp, c_fp, C.c_int, c_fp, c_fp, C.c_int, c_ip, c_fp]
    fn.restype = C.c_int
    rc = fn(len(u), fp(u), fp(v), len(qu), fp(qu), fp(qv), k, ip(idx), fp(dist))
    assert rc == 0
    return idx, dist


def kdtree_knn(u, v, qu, qv, k, kind="f32"):
    """the oracle's restatement of nanoflann's tree build + KNN search (tie order included)"""
    return _knn(lib(kind).orc_kdtree_knn, u, v, qu, qv, k)


_REF_NANOFLANN = None


def ref_nanoflann():
    """oracle/_ref/libref_nanoflann.so = the REFERENCE's own util/nanoflann.h behind our driver (oracle/ref_nanoflann.cpp); built here by `make -C oracle`
    while /root/reference exists, shipped prebuilt to the GPU box. None if it was never built."""
    global _REF_NANOFLANN
    if _REF_NANOFLANN is None:
        path = os.path.join(_DIR, "_ref", "libref_nanoflann.so")
        if not os.path.exists(path):
            build()
        _REF_NANOFLANN = C.CDLL(path) if os.path.exists(path) else False
    return _REF_NANOFLANN or None


def ref_nanoflann_knn(u, v, qu, qv, k):
    return _knn(ref_nanoflann().ref_nanoflann_knn, u, v, qu, qv, k)


def grid_max_selection(dI3, w, h, pot, THFac=1.0, kind="f32"):
    m = np.zeros((h, w), np.uint8)
    L = lib(kind)
    L.orc_grid_max_selection.argtypes = [c_fp, c_u8p, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_grid_max_selection.restype = C.c_int
    n = L.orc_grid_max_selection(fp(np.ascontiguousarray(dI3, np.float32)), u8p(m), w, h, pot, THFac)
    return m, n


def make_pixel_status(dI3, w, h, desiredDensity, sparsityFactor, recsLeft=5, THFac=1.0, kind="f32"):
    """-> (map [h,w] uint8, numGoodPoints, the updated global sparsityFactor)"""
    m, sf = np.zeros((h, w), np.uint8), np.array([sparsityFactor], np.int32)
    L = lib(kind)
    L.orc_make_pixel_status.argtypes = [c_fp, c_u8p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, c_ip]
    L.orc_make_pixel_status.restype = C.c_int
    n = L.orc_make_pixel_status(fp(np.ascontiguousarray(dI3, np.float32)), u8p(m), w, h, desiredDensity, recsLeft, THFac, ip(sf))
    return m, n, int(sf[0])


class Initializer:
    """CoarseInitializer (setFirst + trackFrame and everything they call) on the CPU."""
    FIELDS = {"u": (np.float32, 1), "v": (np.float32, 1), "idepth": (np.float32, 1), "idepth_new": (np.float32, 1), "iR": (np.float32, 1), "lastHessian": (np.float32, 1),
              "energy": (np.float32, 2), "outlierTH": (np.float32, 1), "my_type": (np.float32, 1), "neighboursDist": (np.float32, 10), "parentDist": (np.float32, 1),
              "isGood": (np.uint8, 1), "parent": (np.int32, 1), "neighbours": (np.int32, 10), "maxstep": (np.float32, 1),
              "lastHessian_new": (np.float32, 1), "energy_new": (np.float32, 2), "isGood_new": (np.uint8, 1), "iRSumNum": (np.float32, 1)}

    def __init__(self, w, h, levels, K, kind="f32"):
        self.L = lib(kind)
        self.kind, self.w, self.h, self.levels = kind, w, h, levels
        self.L.orc_initf_create.restype = C.c_void_p
        self.L.orc_initf_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        self.L.orc_initf_destroy.argtypes = [C.c_void_p]
        self.L.orc_initf_set_first.argtypes = [C.c_void_p, C.POINTER(c_fp), c_fp, c_ip]
        self.L.orc_initf_track_frame.argtypes = [C.c_void_p, C.POINTER(c_fp), C.c_float, C.c_float]
        self.L.orc_initf_track_frame.restype = C.c_int
        self.L.orc_initf_num.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_initf_num.restype = C.c_int
        self.L.orc_initf_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
        self.L.orc_initf_get.restype = C.c_int
        self.L.orc_initf_get_state.argtypes = [C.c_void_p, c_dp, c_dp, c_ip]
        self.h_ = self.L.orc_initf_create(w, h, levels, *[float(x) for x in K])

    def close(self):
        if self.h_:
            self.L.orc_initf_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        self.close()

    def _pyr(self, img):
        dI, ab = make_images(img, self.levels, self.kind)
        offs = [self.L.orc_pyr_offset(self.w, self.h, l) for l in range(self.levels + 1)]
        lv = [np.ascontiguousarray(dI[offs[l]:offs[l + 1]]) for l in range(self.levels)]
        return lv, dI, ab, offs

    def set_first(self, img, randomPattern, sparsityFactor=5):
        """-> the updated global sparsityFactor. Level 0 is selected by a fresh PixelSelector (currentPotential 3): makeMaps(., 0.03*w*h, 1, false, 2)."""
        lv, dI, ab, offs = self._pyr(img)
        w, h = self.w, self.h
        ag = [ab[offs[l]:offs[l + 1]] for l in range(3)]
        m0, _, _ = pixsel_make_maps(lv[0], ag[0], ag[1], ag[2], w, h, randomPattern, np.float32(0.03) * w * h, 3, recursionsLeft=1, thFactor=2.0, kind=self.kind)
        self.status0 = m0
        arr = (c_fp * self.levels)(*[fp(x) for x in lv])
        sf = np.array([sparsityFactor], np.int32)
        self.L.orc_initf_set_first(self.h_, arr, fp(np.ascontiguousarray(m0, np.float32)), ip(sf))
        return int(sf[0])

    def track_frame(self, img, exposure_first=1.0, exposure_new=1.0):
        lv, _, _, _ = self._pyr(img)
        arr = (c_fp * self.levels)(*[fp(x) for x in lv])
        return bool(self.L.orc_initf_track_frame(self.h_, arr, exposure_first, exposure_new))

    def num(self, lvl):
        return self.L.orc_initf_num(self.h_, lvl)

    SWEEPS = dict(opt_reg=0, propagate_up=1, propagate_down=2, reset_points=3)

    def sweep(self, which, lvl):
        """one of trackFrame's sweeps alone: optReg(lvl), propagateUp(srcLvl), propagateDown(srcLvl), resetPoints(lvl)"""
        self.L.orc_initf_sweep.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.orc_initf_sweep.restype = None
        self.L.orc_initf_sweep(self.h_, self.SWEEPS[which], lvl)

    def get(self, lvl, field):
        dt, m = self.FIELDS[field]
        n = self.num(lvl)
        out = np.zeros((n, m) if m > 1 else n, dt)
        assert self.L.orc_initf_get(self.h_, lvl, field.encode(), out.ctypes.data_as(C.c_void_p)) == 0
        return out

    CARRIED = ("idepth", "idepth_new", "iR", "lastHessian", "energy", "isGood", "maxstep", "lastHessian_new", "energy_new", "isGood_new", "iRSumNum")

    def set(self, lvl, field, value):
        dt, m = self.FIELDS[field]
        a = np.ascontiguousarray(value, dt)
        assert a.size == self.num(lvl) * m
        self.L.orc_initf_set.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
        assert self.L.orc_initf_set(self.h_, lvl, field.encode(), a.ctypes.data_as(C.c_void_p)) == 0

    def carried(self):
        """everything trackFrame reads from the previous frame: the pose / affine / snap state and, per level, the CARRIED Pnt members"""
        return dict(state=self.state(), points=[{k: self.get(l, k) for k in self.CARRIED} for l in range(self.levels)])

    def set_carried(self, car):
        s = car["state"]
        self.L.orc_initf_set_state.argtypes = [C.c_void_p, c_dp, c_dp, c_ip]
        self.L.orc_initf_set_state(self.h_, dp(np.ascontiguousarray(s["thisToNext"], np.float64).reshape(-1)), dp(np.ascontiguousarray(s["aff"], np.float64)),
                                   ip(np.array([int(s["snapped"]), s["frameID"], s["snappedAt"]], np.int32)))
        for l in range(self.levels):
            for k in self.CARRIED:
                self.set(l, k, car["points"][l][k])

    def state(self):
        T, aff, st = np.zeros(12), np.zeros(2), np.zeros(4, np.int32)
        self.L.orc_initf_get_state(self.h_, dp(T), dp(aff), ip(st))
        return dict(thisToNext=T.reshape(3, 4), aff=aff, snapped=bool(st[0]), frameID=int(st[1]), snappedAt=int(st[2]), n_evals=int(st[3]))


# ------------------------------------------------------------------ image side of the dataset reader (orc_undist.c)
def undistort(raw, G, vinv, photometric, factor, remapX, remapY, w, h, kind="f32"):
    """Undistort::undistort<T>: raw [hOrg, wOrg] uint8 / uint16 -> float32 [h, w]"""
    raw = np.ascontiguousarray(raw)
    hOrg, wOrg = raw.shape
    out = np.zeros((h, w), np.float32)
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    G, vinv, remapX, remapY = f(G), f(vinv), f(remapX), f(remapY)
    L = lib(kind)
    L.orc_undistort.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_fp, c_fp, C.c_int, C.c_float, c_fp, c_fp, C.c_int, C.c_int, c_fp]
    L.orc_undistort(raw.ctypes.data_as(C.c_void_p), raw.dtype.itemsize, wOrg, hOrg, fp(G), fp(vinv), int(photometric), float(factor), fp(remapX), fp(remapY), w, h, fp(out))
    return out


def resize_nearest_u8(src, w, h, kind="f32"):
    src = np.ascontiguousarray(src, np.uint8)
    hOrg, wOrg = src.shape[:2]
    ch = 1 if src.ndim == 2 else src.shape[2]
    dst = np.zeros((h, w) if src.ndim == 2 else (h, w, ch), np.uint8)
    L = lib(kind)
    L.orc_resize_nearest_u8.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, c_u8p, C.c_int, C.c_int]
    L.orc_resize_nearest_u8(u8p(src), wOrg, hOrg, ch, u8p(dst), w, h)
    return dst
