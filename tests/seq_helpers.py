"""A multi-keyframe synthetic sequence driven through the hot path: the control plane of FullSystem::makeKeyFrame (FullSystem.cpp:1279-1564) restated
as TEST INFRASTRUCTURE, with two interchangeable numeric back-ends — the HIP library through its C-ABI (GpuBackend) and the CPU oracle
(OracleBackend). Per keyframe, as the reference does:

    trackNewestCoarse of the frames since the last keyframe        (CoarseTracker.cpp:1073-1259, FullSystem.cpp:594-597)
    flagFramesForMarginalization                                   (FullSystemMarginalize.cpp:57-143)
    insertFrame (HM/bM grow by a zero block), residuals of the old points towards the new frame, activation of new points
    optimize(6)                                                    (FullSystemOptimize.cpp:398-602)
    removeOutliers, setCoarseTrackingRef                           (FullSystem.cpp:1404)
    flagPointsForRemoval -> marginalizePointsF                     (FullSystem.cpp:930-1026, EnergyFunctional.cpp:615-676)
    marginalizeFrame                                               (EnergyFunctional.cpp:498-610)

STRUCTURE (which frames, which points, which residuals exist, what gets marginalised) is decided ONCE per keyframe from the oracle's state and applied
to both back-ends, so their inputs stay comparable; every NUMBER (poses, affine parameters, inverse depths, calibration, HM/bM, energy thresholds,
tracked poses) is each back-end's own and is carried from keyframe to keyframe (closed loop)."""
import dataclasses

import numpy as np

import orc
from nalo_slam_amd import binding, synth

MAX_FRAMES = 7                # setting_maxFrames (util/settings.cpp:88)
MIN_FRAMES = 5
MIN_POINTS_REMAINING = 0.05   # setting_minPointsRemaining
MIN_IDEPTH_H_MARG = 50.0      # setting_minIdepthH_marg (settings.cpp:77)


def make_sequence(w=640, h=480, n_kf=12, per_kf=350, stride=2, seed=31, step_z=0.4, yaw_deg=0.3, f=None):
    """n_kf keyframes, every `stride`-th frame of a forward-moving camera; per_kf candidate points hosted on every keyframe"""
    F = (n_kf - 1) * stride + 1
    win = synth.make_window(w=w, h=h, W=F, P=per_kf * F, seed=seed, n_extra=0, step_z=step_z, yaw_deg=yaw_deg, f=f)
    return win, [i * stride for i in range(n_kf)]


@dataclasses.dataclass
class FrameNum:               # the numeric state of one window frame, owned by a back-end
    fid: int
    frame_id: int
    evalPT: np.ndarray
    state: np.ndarray
    state_zero: np.ndarray
    th: float
    w2c: np.ndarray


class GpuBackend:
    name = "gpu"

    def __init__(self, win):
        self.win = win
        self.c = binding.Context(win.w, win.h, win.K, n_slots=len(win.images))
        for i in range(len(win.images)):
            self.c.frame_upload(i, win.images[i])
        self.K0 = np.asarray(win.K, np.float64)

    def close(self):
        self.c.close()

    def set_window(self, frames, calib, HM, bM):
        self.c.ba_set_window([f.fid for f in frames], [f.evalPT for f in frames], th=[f.th for f in frames], frame_ids=[f.frame_id for f in frames],
                             states=[f.state for f in frames], states_zero=[f.state_zero for f in frames], calib=calib, calib_zero=self.K0)
        self.c.ba_set_prior(HM, bM)

    def set_points(self, host_idx, u, v, idepth, color, weights, has_prior, exists):
        self.c.ba_set_points(host_idx, u, v, idepth, color, weights, has_prior=has_prior)
        self.c.ba_set_residuals(exists)

    def optimize(self):
        return self.c.ba_optimize(6)

    def read_window(self, frames):
        arr, w2c, cal = self.c.ba_get_frames()
        out = [FrameNum(f.fid, f.frame_id, np.array(a.worldToCam_evalPT).reshape(3, 4), np.array(a.state), np.array(a.state_zero), float(a.frameEnergyTH), w2c[i].copy())
               for i, (f, a) in enumerate(zip(frames, arr))]
        return out, cal

    def read_points(self):
        p = self.c.ba_get_points()
        st, ac, _, _, cp = self.c.ba_get_residuals()
        return p["idepth"], p["HdiF"], st, cp

    def marginalize_points(self, flags):
        self.c.ba_marginalize_points(flags)

    def marginalize_frame(self, idx):
        self.c.ba_marginalize_frame(idx)
        return self.c.ba_get_prior()

    def get_prior(self):
        return self.c.ba_get_prior()

    def set_tracking_ref(self, fid, calib, Ku, Kv, nid, hdi):
        self.c._ck(self.c.L.nalo_trk_make_k(self.c.h_, *[float(x) for x in calib]))
        self.c.trk_set_ref(fid, Ku, Kv, nid, hdi)

    def track(self, fid_new, T0, aff0, ref_aff):
        ok, T, aff, lr, lf, _ = self.c.trk_track(fid_new, T0, aff0, ref_aff, [1.0, 1.0], self.c.levels - 1)
        return ok, T, aff


class OracleBackend:
    name = "oracle"

    def __init__(self, win, kind="f32"):
        self.win, self.kind = win, kind
        orc.lib(kind).orc_set_sum_mode(0)
        self.dI = [orc.make_images(win.images[i], win.levels, kind)[0] for i in range(len(win.images))]
        self.n0 = win.w * win.h
        self.K0 = np.asarray(win.K, np.float64)
        self.ba = None
        self.trk = None

    def close(self):
        self.ba = None

    def set_window(self, frames, calib, HM, bM):
        self._frames, self._calib, self._HM, self._bM = frames, np.asarray(calib, np.float64), HM, bM

    def set_points(self, host_idx, u, v, idepth, color, weights, has_prior, exists):
        fr, W = self._frames, len(self._frames)
        ba = orc.BA(W, len(host_idx), self.win.w, self.win.h, tuple(self._calib), self.kind)
        for i, f in enumerate(fr):
            ba.set_frame_full(i, self.dI[f.fid][:self.n0], f.evalPT, f.state, f.state_zero, 1.0, f.th, f.frame_id)
        ba.set_points(host_idx, u, v, idepth, color, weights, has_prior=has_prior)
        ba.set_calib_zero(self.K0)
        ba.set_residuals(exists)
        ba.set_prior(self._HM, self._bM)
        ba.prepare()
        self.ba = ba

    def optimize(self):
        return self.ba.optimize(6)

    def read_window(self, frames):
        out = []
        for i, f in enumerate(frames):
            o = self.ba.frame(i)
            out.append(FrameNum(f.fid, f.frame_id, o["evalPT"].copy(), o["state"].copy(), o["state_zero"].copy(), float(o["frameEnergyTH"]), o["worldToCam"].copy()))
        return out, self.ba.calib()

    def read_points(self):
        p = self.ba.points()
        st, ac, _, _ = self.ba.slots()
        return p["idepth"], p["HdiF"], st, self.ba.center_projected()

    def marginalize_points(self, flags):
        self.ba.marginalize_points(flags)

    def marginalize_frame(self, idx):
        return self.ba.marginalize_frame(idx)

    def get_prior(self):
        return self.ba.get_prior()

    def set_tracking_ref(self, fid, calib, Ku, Kv, nid, hdi):
        self.trk = orc.Tracker(self.win.w, self.win.h, self.win.levels, tuple(float(x) for x in calib), self.kind)
        self.trk.set_ref(self.dI[fid], Ku, Kv, nid, hdi)

    def track(self, fid_new, T0, aff0, ref_aff):
        ok, T, aff, lr, lf = self.trk.track(self.dI[fid_new], T0, aff0, ref_aff, [1.0, 1.0], self.win.levels - 1)
        return ok, T, aff


def _se3_exp(xi):
    return orc.se3_exp(np.asarray(xi, np.float64))


class SequenceDriver:
    """runs the same keyframe sequence on several back-ends in lock-step; structure from backends[0] (the oracle), numbers per back-end"""

    def __init__(self, win, kf_fids, backends, teacher=False):
        """teacher = True: before every keyframe the other back-ends are handed back-end 0's NUMBERS (frames, calibration, prior, inverse depths,
        tracker affine) — a per-keyframe comparison on identical carried state, free of closed-loop drift; False: every back-end keeps its own"""
        self.win, self.kf, self.B, self.teacher = win, kf_fids, backends, teacher
        nb = len(backends)
        self.frames = [[] for _ in range(nb)]          # per back-end: list[FrameNum] of the current window
        self.calib = [np.asarray(win.K, np.float64) for _ in range(nb)]
        self.prior = [(np.zeros((4, 4)), np.zeros(4)) for _ in range(nb)]
        self.idepth = [win.idepth.copy() for _ in range(nb)]      # per back-end inverse depths of the whole candidate pool
        P = len(win.host)
        self.active = np.zeros(P, bool)
        self.has_prior = np.zeros(P, np.int32)
        self.exists = np.zeros((P, len(win.images)), bool)        # residual (point, target fid)
        self.n_marg = {f: 0 for f in kf_fids}                     # per host frame: points marginalised / dropped (flagFramesForMarginalization's `out`)
        self.n_out = {f: 0 for f in kf_fids}
        self.ref_aff = [np.zeros(2) for _ in range(nb)]
        self.log = []

    # ---- helpers
    def _win_fids(self):
        return [f.fid for f in self.frames[0]]

    def _submit(self, b, new_frame=None):
        """set_window (+ insertFrame of new_frame), set_points, set_residuals on back-end b for the current structure"""
        fr = list(self.frames[b]) + ([new_frame] if new_frame is not None else [])
        HM, bM = self.prior[b]
        n = 8 * len(fr) + 4
        if HM.shape[0] == n - 8:                                  # EnergyFunctional::insertFrame: zero block for the new frame
            H2, b2 = np.zeros((n, n)), np.zeros(n)
            H2[:n - 8, :n - 8], b2[:n - 8] = HM, bM
            HM, bM = H2, b2
        assert HM.shape[0] == n
        self.B[b].set_window(fr, self.calib[b], HM, bM)
        fids = [f.fid for f in fr]
        ids = np.nonzero(self.active)[0]
        host_idx = np.array([fids.index(h) for h in self.win.host[ids]], np.int32)
        ex = self.exists[ids][:, fids].astype(np.uint8)
        self.B[b].set_points(host_idx, self.win.u[ids], self.win.v[ids], self.idepth[b][ids], self.win.color[ids], self.win.weights[ids], self.has_prior[ids], ex)
        self.frames[b] = fr
        return ids

    def _flag_frames(self, latest_frame_id):
        """FullSystemMarginalize.cpp:57-143 on the oracle's window (called before the new frame is pushed)"""
        fr = self.frames[0]
        flagged = []
        for f in fr:
            n_in = int((self.active & (self.win.host == f.fid)).sum()) + (350 if f.fid == fr[-1].fid else 0)     # immature points of the newest frame
            n_out = self.n_marg[f.fid] + self.n_out[f.fid]
            if n_in < MIN_POINTS_REMAINING * (n_in + n_out) and len(fr) - len(flagged) > MIN_FRAMES:
                flagged.append(f.fid)
        if len(fr) - len(flagged) >= MAX_FRAMES:
            best, best_fid = 1.0, None
            c2w = {f.fid: synth.se3_inv(f.w2c) for f in fr}
            for f in fr:
                if f.frame_id > latest_frame_id - 1 or f.frame_id == 0:
                    continue
                score = 0.0
                for g in fr:
                    if g.frame_id > latest_frame_id - 1 + 1 or g.fid == f.fid:
                        continue
                    d = np.linalg.norm(synth.se3_mul(g.w2c, c2w[f.fid])[:, 3])
                    score += 1.0 / (1e-5 + d)
                score *= -np.sqrt(np.linalg.norm(synth.se3_mul(fr[-1].w2c, c2w[f.fid])[:, 3]))
                if score < best:
                    best, best_fid = score, f.fid
            if best_fid is not None and best_fid not in flagged:
                flagged.append(best_fid)
        return flagged

    # ---- the sequence
    def bootstrap(self):
        """window = {KF0, KF1}: KF0 at the true pose (gauge), KF1 slightly off; KF0's candidate points active with a depth prior (initializeFromInitializer
        sets hasDepthPrior, FullSystem.cpp:1630-1650)"""
        win, (f0, f1) = self.win, self.kf[:2]
        rng = np.random.RandomState(11)
        off = np.r_[0.01 * rng.randn(3), 0.001 * rng.randn(3)]
        for b in range(len(self.B)):
            self.frames[b] = [FrameNum(f0, 0, win.world_to_cam[f0].copy(), np.zeros(10), np.zeros(10), 512.0, win.world_to_cam[f0].copy())]
            self.prior[b] = (np.zeros((12, 12)), np.zeros(12))
        ids = np.nonzero(win.host == f0)[0]
        self.active[ids] = True
        self.has_prior[ids] = 1
        self.exists[ids, f1] = True
        ev1 = synth.se3_mul(_se3_exp(off), win.world_to_cam[f1])
        new = FrameNum(f1, 1, ev1, np.zeros(10), np.zeros(10), 512.0, ev1)
        return self._keyframe_tail(1, [dataclasses.replace(new) for _ in self.B], flagged=[])

    def add_keyframe(self, k):
        """keyframe number k >= 2: track the frames since the last keyframe, then the back-end part of makeKeyFrame"""
        win = self.win
        ref_fid, new_fid = self.kf[k - 1], self.kf[k]
        if self.teacher:
            for b in range(1, len(self.B)):
                self.frames[b] = [dataclasses.replace(f) for f in self.frames[0]]
                self.calib[b] = self.calib[0].copy()
                self.prior[b] = (self.prior[0][0].copy(), self.prior[0][1].copy())
                self.idepth[b] = self.idepth[0].copy()
                self.ref_aff[b] = self.ref_aff[0].copy()
        tracked = {}
        new_frames = []
        for b, be in enumerate(self.B):
            ref = self.frames[b][-1]
            assert ref.fid == ref_fid
            aff_prev = self.ref_aff[b].copy()                              # the reference starts from the last frame's aff_g2l (FullSystem.cpp:529-537)
            for fid in range(ref_fid + 1, new_fid + 1):
                T_true = synth.se3_mul(win.world_to_cam[fid], synth.se3_inv(win.world_to_cam[ref_fid]))
                T0 = _se3_exp(orc.se3_log(T_true) * 0.9)                   # motion-model stand-in: the same initial guess on every back-end
                ok, T, aff = be.track(fid, T0, aff_prev, self.ref_aff[b])
                tracked[(b, fid)] = (ok, T, aff)
                aff_prev = aff
            ok, T, aff = tracked[(b, new_fid)]
            ev = synth.se3_mul(T, ref.w2c)                                 # camToWorld = trackingRef.camToWorld * camToTrackingRef (FullSystem.cpp:1285)
            st = np.zeros(10)
            st[6], st[7] = np.float32(1.0 / 10.0) * aff[0], np.float32(1.0 / 1000.0) * aff[1]      # setEvalPT_scaled (HessianBlocks.h:247-255)
            new_frames.append(FrameNum(new_fid, k, ev, st.copy(), st.copy(), 512.0, ev))
        flagged = self._flag_frames(self.frames[0][-1].frame_id)
        # residuals of the old points towards the new frame (FullSystem.cpp:1335-1348), activation of the previous keyframe's candidates
        ids_old = np.nonzero(self.active)[0]
        self.exists[ids_old, new_fid] = True
        ids_new = np.nonzero(self.win.host == ref_fid)[0]
        self.active[ids_new] = True
        for f in self._win_fids() + [new_fid]:
            if f != ref_fid:
                self.exists[ids_new, f] = True
        out = self._keyframe_tail(k, new_frames, flagged)
        out["tracked"] = tracked
        return out

    def _keyframe_tail(self, k, new_frames, flagged):
        win, nb = self.win, len(self.B)
        ids = None
        rmse = []
        for b in range(nb):
            ids = self._submit(b, new_frames[b])
            rmse.append(self.B[b].optimize())
        fids = self._win_fids()
        W = len(fids)
        res_state, cpt, hdi = [], [], []
        for b, be in enumerate(self.B):
            self.frames[b], self.calib[b] = be.read_window(self.frames[b])
            idp, h, st, cp = be.read_points()
            self.idepth[b][ids] = idp
            res_state.append(st); cpt.append(cp); hdi.append(h)
        # structure from the oracle: surviving residuals, removeOutliers (points without residuals)
        st0 = res_state[0]
        self.exists[np.ix_(ids, fids)] = st0 >= 0
        dead = ids[(st0 >= 0).sum(1) == 0]
        for p in dead:
            self.n_out[win.host[p]] += 1
        self.active[dead] = False
        # setCoarseTrackingRef: IN residuals that target the newest keyframe (each back-end's own numbers)
        for b, be in enumerate(self.B):
            m = res_state[b][:, W - 1] == 0
            be.set_tracking_ref(fids[-1], self.calib[b], cpt[b][m, W - 1, 0], cpt[b][m, W - 1, 1], cpt[b][m, W - 1, 2], hdi[b][m])
            nf = self.frames[b][-1]
            self.ref_aff[b] = np.array([nf.state[6] * 10.0, nf.state[7] * 1000.0])          # aff_g2l = state_scaled[6:8]
        # flagPointsForRemoval (simplified isOOB / isInlierNew, decided on the oracle's state)
        alive = (st0 >= 0).sum(1) > 0
        nres = (st0 >= 0).sum(1)
        host_flagged = np.isin(win.host[ids], flagged)
        lost_newest = (st0[:, W - 1] < 0) & (win.host[ids] != fids[-1])
        vis_in_marg = (st0[:, [fids.index(f) for f in flagged]] >= 0).sum(1) if flagged else np.zeros(len(ids), int)
        oob = alive & (host_flagged | lost_newest | ((nres >= 3) & (nres - vis_in_marg < 3)))
        Hdd = 1.0 / np.maximum(hdi[0], 1e-30)
        marg = oob & (nres >= 3) & (Hdd > MIN_IDEPTH_H_MARG) & (self.idepth[0][ids] > 0)
        drop = oob & ~marg
        flags = marg.astype(np.uint8)
        for b, be in enumerate(self.B):
            be.marginalize_points(flags)
            self.prior[b] = be.get_prior()
        for p in ids[marg]:
            self.n_marg[win.host[p]] += 1
        for p in ids[drop]:
            self.n_out[win.host[p]] += 1
        self.active[ids[marg | drop]] = False
        # marginalizeFrame for every flagged frame (their residual columns go with them)
        for fid in flagged:
            for b, be in enumerate(self.B):
                idx = [f.fid for f in self.frames[b]].index(fid)
                self._resubmit_for_frame_marg(b)
                self.prior[b] = be.marginalize_frame(idx)
                self.frames[b] = [f for f in self.frames[b] if f.fid != fid]
            self.exists[:, fid] = False
        rec = dict(k=k, fids=fids, flagged=flagged, n_active=int(self.active.sum()), n_marg=int(marg.sum()), n_drop=int(drop.sum() + len(dead)), rmse=rmse,
                   frames=[list(fr) for fr in self.frames], prior=[(H.copy(), bb.copy()) for H, bb in self.prior], calib=[c.copy() for c in self.calib],
                   idepth=[self.idepth[b][self.active].copy() for b in range(nb)],
                   state_mismatch=[int((res_state[b] != st0).sum()) for b in range(nb)], n_res=int((st0 >= 0).sum()))
        self.log.append(rec)
        return rec

    def _resubmit_for_frame_marg(self, b):
        """the frame being marginalised must not host active points any more: hand the surviving structure to the back-end first (the oracle needs a BA
        object of the current window to run marginalizeFrame on; the library checks the 'no points left' assertion of EnergyFunctional.cpp:505)"""
        self._submit(b)
