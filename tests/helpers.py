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
