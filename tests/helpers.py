"""Shared helpers for the parity tests (oracle = checker only)."""
import numpy as np

import orc
from nalo_slam_amd import synth


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b).max()
    s = max(np.abs(b).max(), 1e-300)
    return d / s


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
