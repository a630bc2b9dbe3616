"""Seeded synthetic windows for the hot path: analytic textured scene -> keyframes,
active points (host, u, v, idepth, color[8], weights[8]) and the residual graph.

Scene: ground plane y = 1.6 m (y down, camera looks along +z) and a fronto-parallel wall at z = wall_z.
Albedo = 128 + amp * sum_k sin(f_k a + phi_k) cos(g_k b + psi_k), clamped to [5, 250], with (a, b) the
in-plane coordinates. Everything is generated on the host with numpy; no files.

Point attributes follow ImmaturePoint::ImmaturePoint (reference src/FullSystem/ImmaturePoint.cpp:33-54):
integer pixel position, color[idx] = I(u+dx, v+dy), weights[idx] = sqrt(c / (c + |grad|^2)), c = 50^2.
"""
from __future__ import annotations

import dataclasses
import numpy as np

PATTERN = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], dtype=np.int32)
OUTLIER_TH_SUMCOMP = 50.0 * 50.0


def pyr_levels(w: int, h: int, max_levels: int = 6) -> int:
    """util/globalCalib.cpp:50-55 level rule."""
    lv, wl, hl = 1, w, h
    while wl % 2 == 0 and hl % 2 == 0 and wl * hl > 5000 and lv < max_levels:
        wl //= 2
        hl //= 2
        lv += 1
    return lv


def so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def se3_inv(T):
    R, t = T[:, :3], T[:, 3]
    return np.concatenate([R.T, (-R.T @ t)[:, None]], axis=1)


def se3_mul(A, B):
    return np.concatenate([A[:, :3] @ B[:, :3], (A[:, :3] @ B[:, 3] + A[:, 3])[:, None]], axis=1)


@dataclasses.dataclass
class Window:
    w: int
    h: int
    levels: int
    K: tuple                     # fx, fy, cx, cy
    images: np.ndarray           # [F, h, w] float32 irradiance (F = W keyframes + n_extra tracked frames)
    depth: np.ndarray            # [F, h, w] float32 true depth
    world_to_cam: np.ndarray     # [F, 3, 4] float64 ground truth
    W: int
    host: np.ndarray             # [P] int32
    u: np.ndarray                # [P] float32 (integer valued)
    v: np.ndarray
    idepth: np.ndarray           # [P] float32 (noisy)
    idepth_true: np.ndarray
    color: np.ndarray            # [P, 8] float32
    weights: np.ndarray          # [P, 8] float32
    exists: np.ndarray           # [P, W] uint8 residual graph


class Scene:
    def __init__(self, seed: int = 20240601, wall_z: float = 40.0, ground_y: float = 1.6, amp: float = 50.0, freq_scale: float = 1.0):
        rng = np.random.RandomState(seed)
        self.wall_z, self.ground_y, self.amp = wall_z, ground_y, amp
        self.fg = rng.uniform(2.0, 14.0, size=(6, 2)) * freq_scale     # ground freqs (rad/m)
        self.pg = rng.uniform(0, 2 * np.pi, size=(6, 2))
        self.fw = rng.uniform(1.0, 8.0, size=(6, 2)) * freq_scale      # wall freqs
        self.pw = rng.uniform(0, 2 * np.pi, size=(6, 2))

    def _tex(self, a, b, f, p):
        acc = np.zeros_like(a)
        for k in range(f.shape[0]):
            acc += np.sin(f[k, 0] * a + p[k, 0]) * np.cos(f[k, 1] * b + p[k, 1])
        return acc

    def render(self, w, h, K, world_to_cam):
        fx, fy, cx, cy = K
        c2w = se3_inv(world_to_cam)
        uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        d = np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], axis=-1)
        dw = d @ c2w[:, :3].T
        o = c2w[:, 3]
        with np.errstate(divide="ignore", invalid="ignore"):
            sg = np.where(dw[..., 1] > 1e-9, (self.ground_y - o[1]) / dw[..., 1], np.inf)
            sw = np.where(dw[..., 2] > 1e-9, (self.wall_z - o[2]) / dw[..., 2], np.inf)
        sg = np.where(sg > 0, sg, np.inf)
        sw = np.where(sw > 0, sw, np.inf)
        s = np.minimum(sg, sw)
        Pw = o[None, None, :] + s[..., None] * dw
        on_ground = sg <= sw
        tg = self._tex(Pw[..., 0], Pw[..., 2], self.fg, self.pg)
        tw = self._tex(Pw[..., 0], Pw[..., 1], self.fw, self.pw)
        img = 128.0 + self.amp * np.where(on_ground, tg, tw) / 2.0
        return np.clip(img, 5.0, 250.0).astype(np.float32), s.astype(np.float32)


def make_window(w=640, h=480, W=8, P=2000, seed=7, n_extra=1, idepth_noise=0.01, f=None,
                step_z=0.8, yaw_deg=0.5, full_graph=True, pose_seed_scene=20240601, min_grad2=50.0, freq_scale=1.0) -> Window:
    f = float(f if f is not None else 0.52 * w)
    K = (f, f, (w - 1) / 2.0, (h - 1) / 2.0)
    scene = Scene(pose_seed_scene, freq_scale=freq_scale)
    F = W + n_extra
    w2c = np.zeros((F, 3, 4))
    for i in range(F):
        R_c2w = so3_exp(np.array([0.0, np.deg2rad(yaw_deg) * i, 0.0]))
        t_c2w = np.array([0.03 * i, 0.0, step_z * i])
        c2w = np.concatenate([R_c2w, t_c2w[:, None]], axis=1)
        w2c[i] = se3_inv(c2w)
    imgs = np.zeros((F, h, w), np.float32)
    depth = np.zeros((F, h, w), np.float32)
    for i in range(F):
        imgs[i], depth[i] = scene.render(w, h, K, w2c[i])
    rng = np.random.RandomState(seed)
    host = (np.arange(P) % W).astype(np.int32)
    u = np.zeros(P, np.int64)
    v = np.zeros(P, np.int64)
    todo = np.arange(P)
    gx = np.zeros_like(imgs)
    gy = np.zeros_like(imgs)
    gx[:, :, 1:-1] = 0.5 * (imgs[:, :, 2:] - imgs[:, :, :-2])
    gy[:, 1:-1, :] = 0.5 * (imgs[:, 2:, :] - imgs[:, :-2, :])
    for _ in range(200):
        if todo.size == 0:
            break
        uu = rng.randint(4, w - 4, size=todo.size)
        vv = rng.randint(4, h - 4, size=todo.size)
        g2 = gx[host[todo], vv, uu] ** 2 + gy[host[todo], vv, uu] ** 2
        ok = (g2 >= min_grad2) & np.isfinite(depth[host[todo], vv, uu])
        u[todo[ok]] = uu[ok]
        v[todo[ok]] = vv[ok]
        todo = todo[~ok]
    if todo.size:
        raise RuntimeError("could not place %d points" % todo.size)
    idt = (1.0 / depth[host, v, u]).astype(np.float32)
    idn = (idt * (1.0 + idepth_noise * rng.randn(P))).astype(np.float32)
    color = np.zeros((P, 8), np.float32)
    wts = np.zeros((P, 8), np.float32)
    for k in range(8):
        uk, vk = u + PATTERN[k, 0], v + PATTERN[k, 1]
        color[:, k] = imgs[host, vk, uk]
        g2 = gx[host, vk, uk] ** 2 + gy[host, vk, uk] ** 2
        wts[:, k] = np.sqrt(OUTLIER_TH_SUMCOMP / (OUTLIER_TH_SUMCOMP + g2)).astype(np.float32)
    exists = np.ones((P, W), np.uint8)
    exists[np.arange(P), host] = 0
    if not full_graph:
        drop = rng.rand(P, W) < 0.3
        exists[drop] = 0
    return Window(w, h, pyr_levels(w, h), K, imgs, depth, w2c, W, host, u.astype(np.float32), v.astype(np.float32),
                  idn, idt, color, wts, exists)


def perturbed_poses(win: Window, seed=3, sigma_t=0.01, sigma_r=0.001):
    """evalPT = truth; returns per-frame unscaled state[0:6] perturbations (frame 0 kept fixed)."""
    rng = np.random.RandomState(seed)
    st = np.zeros((win.W, 6))
    st[1:, :3] = sigma_t * rng.randn(win.W - 1, 3) / 0.5     # state_scaled = 0.5 * state for translation
    st[1:, 3:] = sigma_r * rng.randn(win.W - 1, 3)
    return st


def shard_window(win: Window, rank: int, world: int) -> Window:
    """The part of a window's active points rank `rank` of `world` holds (points keep their residuals, frames are replicated): the partition
    is the library's (nalo_shard_points: the same share of every host frame, a contiguous Hilbert range of the host's points)."""
    if world == 1:
        return win
    from . import binding
    idx = binding.shard_points(win.host, win.u, win.v, win.W, win.w, win.h, rank, world)
    return dataclasses.replace(win, host=win.host[idx], u=win.u[idx], v=win.v[idx], idepth=win.idepth[idx], idepth_true=win.idepth_true[idx],
                               color=win.color[idx], weights=win.weights[idx], exists=win.exists[idx])
