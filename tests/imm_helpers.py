"""Inputs for the immature-point tests (shared by the CPU oracle test and the GPU parity test)."""
import numpy as np

from nalo_slam_amd import synth


def imm_points(win, per_host=1000, seed=1, margin=8):
    """integer pixel positions (PixelSelector output) for every host frame of the window -> u, v (int32), host (int32)"""
    rng = np.random.RandomState(seed)
    u, v, host = [], [], []
    for h in range(win.W):
        u.append(rng.randint(margin, win.w - margin, per_host))
        v.append(rng.randint(margin, win.h - margin, per_host))
        host.append(np.full(per_host, h))
    u, v, host = np.concatenate(u).astype(np.int32), np.concatenate(v).astype(np.int32), np.concatenate(host).astype(np.int32)
    ok = np.isfinite(win.depth[host, v, u]) if isinstance(win.depth, np.ndarray) else np.array([np.isfinite(win.depth[h][y, x]) for h, y, x in zip(host, v, u)])
    return u[ok], v[ok], host[ok]


def host_to_new(win, new):
    """what FullSystem::traceNewCoarse computes per host (FullSystem.cpp:713-721): KRKi [W,9], Kt [W,3], affine pair [W,2] (identity brightness)"""
    fx, fy, cx, cy = win.K
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    Ki = np.linalg.inv(K)
    KRKi, Kt, aff = [], [], []
    for h in range(win.W):
        T = synth.se3_mul(win.world_to_cam[new], synth.se3_inv(win.world_to_cam[h]))
        KRKi.append((K @ T[:, :3] @ Ki).reshape(-1)); Kt.append(K @ T[:, 3]); aff.append([1.0, 0.0])
    return np.asarray(KRKi, np.float32), np.asarray(Kt, np.float32), np.asarray(aff, np.float32)


def true_idepth(win, u, v, host):
    return np.array([1.0 / win.depth[h][y, x] for h, y, x in zip(host, v, u)], np.float32)
