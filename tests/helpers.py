"""Shared helpers for the parity tests (oracle = checker only)."""
import numpy as np

import orc
from nalo_slam_amd import synth


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b).max()
    s = max(np.abs(b).max(), 1e-300)
    return d / s


def assert_fp32_faithful(gpu, o32, o64, factor=1.5, floor=1e-7):
    """Per-row check against the ALL-FP64 evaluation of the same inputs: the GPU's fp32 result must be as close to that truth as the reference's own
    fp32 arithmetic (the strict fp32 oracle, o32) is, at the median, the 99 % quantile and the maximum (x factor: FMA contraction / summation order
    differ between the scalar CPU order and the device). Rows are scaled by their own largest fp64 entry."""
    gpu, o32, o64 = (np.asarray(a, np.float64).reshape(len(a), -1) for a in (gpu, o32, o64))
    s = np.maximum(np.abs(o64).max(1), 1e-300)
    mine, ref = np.abs(gpu - o64).max(1) / s, np.abs(o32 - o64).max(1) / s
    for q in (0.5, 0.99, 1.0):
        assert np.quantile(mine, q) <= factor * np.quantile(ref, q) + floor, (q, np.quantile(mine, [0.5, 0.99, 1.0]), np.quantile(ref, [0.5, 0.99, 1.0]))


def tracker_inputs(win, n=3000, seed=1):
    """Residuals targeting the newest keyframe (W-1): centerProjectedTo + HdiF, from the true depth."""
    rng = np.random.RandomState(seed)
    Ku = rng.uniform(5, win.w - 6, n).astype(np.float32)
    Kv = rng.uniform(5, win.h - 6, n).astype(np.float32)
    d = win.depth[win.W - 1][(Kv + 0.5).astype(int), (Ku + 0.5).astype(int)]
    ok = np.isfinite(d)
    Ku, Kv, d = Ku[ok], Kv[ok], d[ok]
    HdiF = (10.0 ** rng.uniform(-6, -3, len(d))).astype(np.float32)
    return Ku, Kv, (1.0 / d).astype(np.float32), HdiF


def true_rel_pose(win, a, b):
    """a -> b"""
    return synth.se3_mul(win.world_to_cam[b], synth.se3_inv(win.world_to_cam[a]))


def pose_dist(A, B):
    return np.linalg.norm(orc.se3_log(synth.se3_mul(A, synth.se3_inv(B))))


def sim3_aligned_dist(A, B):
    """Largest distance between two monocular window trajectories A, B (lists of 3x4 worldToCam) modulo the Sim(3) gauge: poses are taken relative to the
    window's first frame (removes the global SE(3)) and B's relative translations are rescaled by the least-squares scale factor onto A's. (An Umeyama fit
    on the camera centres is degenerate for the forward-moving, nearly collinear trajectories of these sequences.) A monocular photometric window is held
    in place by priors only (frame 0's pose prior, the depth priors, HM): a perturbation — a flipped outlier decision, closed-loop drift — moves it mostly
    ALONG that gauge, which is not an error of the estimate. With fewer than 3 frames the raw distance is returned."""
    if len(A) < 3:
        return max(pose_dist(a, b) for a, b in zip(A, B))
    ra = [synth.se3_mul(a, synth.se3_inv(A[0])) for a in A[1:]]
    rb = [synth.se3_mul(b, synth.se3_inv(B[0])) for b in B[1:]]
    ta, tb = np.array([r[:, 3] for r in ra]), np.array([r[:, 3] for r in rb])
    s = (ta * tb).sum() / max((tb * tb).sum(), 1e-300)
    return max(pose_dist(x, np.c_[y[:, :3], s * y[:, 3]]) for x, y in zip(ra, rb))


_HIP = None


def _hip():
    """the HIP runtime libnalo_gpu.so itself is linked against (ctypes; no torch): device <-> host copies for test hooks that are handed raw device pointers"""
    global _HIP
    if _HIP is None:
        import ctypes
        _HIP = ctypes.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _HIP.hipMemcpy.restype = ctypes.c_int
    return _HIP


def dev_read_f64(ptr, n):
    out = np.zeros(n, np.float64)
    rc = _hip().hipMemcpy(out.ctypes.data, ptr, n * 8, 2)         # hipMemcpyDeviceToHost
    assert rc == 0, rc
    return out


def dev_write_f64(ptr, arr):
    arr = np.ascontiguousarray(arr, np.float64)
    rc = _hip().hipMemcpy(ptr, arr.ctypes.data, arr.size * 8, 1)  # hipMemcpyHostToDevice
    assert rc == 0, rc
