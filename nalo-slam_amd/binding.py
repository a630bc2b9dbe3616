"""ctypes binding of libnalo_gpu.so (the C-ABI in include/nalo_gpu.h). The product path: it fails loudly when
the HIP library is missing or no gfx950 device is present — there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int)
c_u8p = C.POINTER(C.c_uint8)
c_i8p = C.POINTER(C.c_int8)


class FrameState(C.Structure):
    _fields_ = [("slot", C.c_int), ("frame_id", C.c_int), ("worldToCam_evalPT", C.c_double * 12), ("state", C.c_double * 10),
                ("state_zero", C.c_double * 10), ("ab_exposure", C.c_float), ("frameEnergyTH", C.c_float)]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)

EXPORTS = [
    "nalo_create", "nalo_destroy", "nalo_last_error", "nalo_levels", "nalo_sync", "nalo_stream",
    "nalo_frame_upload", "nalo_frame_upload_raw", "nalo_frame_upload_raw_async", "nalo_undist_set", "nalo_frame_upload_async", "nalo_frame_wait", "nalo_host_alloc", "nalo_host_free", "nalo_frame_rebuild", "nalo_frame_download",
    "nalo_trk_make_k", "nalo_trk_set_ref", "nalo_trk_ref_upload", "nalo_trk_set_ref_resident", "nalo_trk_set_pc", "nalo_trk_get_pc", "nalo_trk_append_plane_points", "nalo_trk_get_depth", "nalo_trk_eval", "nalo_trk_track", "nalo_trk_last_evals", "nalo_trk_set_shard",
    "nalo_ba_set_window", "nalo_ba_set_points", "nalo_ba_set_residuals", "nalo_ba_set_prior", "nalo_ba_get_prior",
    "nalo_ba_linearize", "nalo_ba_accumulate", "nalo_ba_accumulate_sc", "nalo_ba_solve_system", "nalo_ba_backup_state",
    "nalo_ba_do_step", "nalo_ba_optimize", "nalo_ba_marginalize_points", "nalo_ba_marginalize_frame", "nalo_ba_set_prior_carry", "nalo_ba_calc_l_energy", "nalo_ba_calc_m_energy", "nalo_ba_plane_scale_fix", "nalo_ba_sw_gray_optimize", "nalo_ba_optimize_stats", "nalo_get_settings", "nalo_set_settings", "nalo_constants", "nalo_constants_device", "nalo_ba_get_frames", "nalo_ba_get_points",
    "nalo_ba_get_residuals", "nalo_ba_get_idepth_zero", "nalo_ba_get_acc13", "nalo_ba_counts", "nalo_ba_set_allreduce", "nalo_ba_set_allreduce_mode", "nalo_ba_set_allreduce_side", "nalo_ba_exchange_failed", "nalo_side_stream", "nalo_rccl_unique_id", "nalo_ba_rccl_init", "nalo_ba_set_rccl_comm", "nalo_ba_rccl_ranks", "nalo_shard_points", "nalo_ba_snapshot", "nalo_ba_restore",
    "nalo_imm_create", "nalo_imm_trace", "nalo_imm_optimize", "nalo_imm_resident_set", "nalo_imm_resident_optimize", "nalo_imm_resident_trace", "nalo_imm_resident_get", "nalo_init_calc_res_and_gs", "nalo_init_do_step", "nalo_init_set_first", "nalo_init_track_frame", "nalo_init_get_state", "nalo_init_get_points", "nalo_init_set_state", "nalo_init_set_points", "nalo_init_get_carried", "nalo_init_sweep", "nalo_dist_make_map", "nalo_pixsel_make_hists",
    "nalo_pixsel_set_random", "nalo_pixsel_select", "nalo_pixsel_make_maps", "nalo_pixsel_make_maps_lidar", "nalo_pixsel_get_selected",
    "nalo_dense_make_map", "nalo_profile_enable", "nalo_profile_select", "nalo_profile_reset", "nalo_profile_get", "nalo_profile_samples", "nalo_profile_sample", "nalo_hbm_calibrate",
]


class Settings(C.Structure):
    """nalo_settings (include/nalo_gpu.h)"""
    _fields_ = [("forceAcceptStep", C.c_int), ("affineOptModeA", C.c_double), ("affineOptModeB", C.c_double), ("minOptIterations", C.c_int)]


def lib_path():
    return os.path.join(_HERE, "libnalo_gpu.so")


def constants():
    """{reference name: value} of the constants the library was compiled with (nalo_constants; needs no device)"""
    L = C.CDLL(lib_path())
    L.nalo_constants.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double)]
    n = L.nalo_constants(0, None, None)
    names, vals = (C.c_char_p * n)(), (C.c_double * n)()
    assert L.nalo_constants(n, names, vals) == n
    return {names[i].decode(): vals[i] for i in range(n)}


def load():
    """dlopen the in-tree library; raises if it has not been built (python nalo-slam_amd/build.py)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise RuntimeError("libnalo_gpu.so is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); no CPU fallback exists")
    L = C.CDLL(p)
    vp = C.c_void_p
    L.nalo_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int]
    L.nalo_destroy.argtypes = [vp]
    L.nalo_destroy.restype = None
    L.nalo_last_error.argtypes = [vp]
    L.nalo_last_error.restype = C.c_char_p
    L.nalo_levels.argtypes = [vp]
    L.nalo_sync.argtypes = [vp]
    L.nalo_stream.argtypes = [vp]
    L.nalo_stream.restype = vp
    L.nalo_frame_upload.argtypes = [vp, C.c_int, c_fp, c_fp, c_u8p, c_fp]
    L.nalo_frame_upload_async.argtypes = [vp, C.c_int, c_fp, c_fp, c_u8p, c_fp]
    L.nalo_frame_wait.argtypes = [vp, C.c_int]
    L.nalo_host_alloc.argtypes = [C.c_size_t]
    L.nalo_host_alloc.restype = vp
    L.nalo_host_free.argtypes = [vp]
    L.nalo_host_free.restype = None
    L.nalo_frame_rebuild.argtypes = [vp, C.c_int]
    L.nalo_frame_download.argtypes = [vp, C.c_int, C.c_int, c_fp, c_fp]
    L.nalo_trk_make_k.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
    L.nalo_trk_set_ref.argtypes = [vp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]
    L.nalo_trk_ref_upload.argtypes = [vp, C.c_int, c_fp, c_fp, c_fp, c_fp]
    L.nalo_trk_set_ref_resident.argtypes = [vp, C.c_int]
    L.nalo_trk_set_pc.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]
    L.nalo_trk_get_pc.argtypes = [vp, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp]
    L.nalo_trk_get_depth.argtypes = [vp, C.c_int, c_fp, c_fp]
    L.nalo_trk_eval.argtypes = [vp, C.c_int, C.c_int, c_dp, c_dp, c_fp, C.c_float, C.c_float, C.c_int, c_dp, c_dp, c_dp]
    L.nalo_trk_track.argtypes = [vp, C.c_int, c_dp, c_dp, c_dp, c_fp, C.c_int, c_dp, c_dp, c_dp, c_ip, c_ip]
    L.nalo_ba_set_window.argtypes = [vp, C.c_int, C.POINTER(FrameState), c_dp, c_dp]
    L.nalo_ba_set_points.argtypes = [vp, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip]
    L.nalo_ba_set_residuals.argtypes = [vp, c_u8p]
    L.nalo_ba_set_prior.argtypes = [vp, c_dp, c_dp]
    L.nalo_ba_get_prior.argtypes = [vp, c_dp, c_dp]
    L.nalo_ba_linearize.argtypes = [vp, C.c_int, c_dp]
    L.nalo_ba_accumulate.argtypes = [vp, C.c_int, c_dp, c_dp]
    L.nalo_ba_accumulate_sc.argtypes = [vp, C.c_int, c_dp, c_dp]
    L.nalo_ba_solve_system.argtypes = [vp, C.c_int, C.c_double, c_dp]
    L.nalo_ba_backup_state.argtypes = [vp]
    L.nalo_ba_do_step.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_ip]
    L.nalo_ba_optimize.argtypes = [vp, C.c_int, C.c_int, c_dp]
    L.nalo_ba_marginalize_points.argtypes = [vp, c_u8p, c_dp, c_dp, c_dp, c_dp]
    L.nalo_ba_marginalize_frame.argtypes = [vp, C.c_int]
    L.nalo_ba_get_frames.argtypes = [vp, C.POINTER(FrameState), c_dp, c_dp]
    L.nalo_ba_get_points.argtypes = [vp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
    L.nalo_ba_get_residuals.argtypes = [vp, c_i8p, c_u8p, c_fp, c_fp, c_fp]
    L.nalo_ba_get_acc13.argtypes = [vp, c_dp]
    L.nalo_ba_counts.argtypes = [vp, c_ip, c_ip, c_ip]
    L.nalo_ba_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.nalo_ba_set_allreduce_mode.argtypes = [vp, C.c_int]
    L.nalo_ba_set_allreduce_side.argtypes = [vp, ALLREDUCE_FN, vp]
    L.nalo_side_stream.argtypes = [vp]
    L.nalo_side_stream.restype = vp
    L.nalo_rccl_unique_id.argtypes = [C.c_char_p]
    L.nalo_ba_rccl_init.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    L.nalo_ba_set_rccl_comm.argtypes = [vp, vp, vp]
    L.nalo_ba_rccl_ranks.argtypes = [vp, c_ip, c_ip]
    L.nalo_shard_points.argtypes = [C.c_int, C.c_int, c_ip, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_ip]
    L.nalo_ba_snapshot.argtypes = [vp]
    L.nalo_init_calc_res_and_gs.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_u8p, c_fp, c_fp, c_dp, c_dp, C.c_float, C.c_float, C.c_float,
                                            c_u8p, c_fp, c_fp, c_fp, c_fp, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.nalo_init_do_step.argtypes = [vp, C.c_int, c_u8p, c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp]
    L.nalo_trk_append_plane_points.argtypes = [vp, c_fp, C.c_float, C.c_int, c_ip, c_ip]
    L.nalo_undist_set.argtypes = [vp, C.c_int, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp, c_fp]
    L.nalo_frame_upload_raw.argtypes = [vp, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, c_u8p, c_u8p, c_fp]
    L.nalo_frame_upload_raw_async.argtypes = [vp, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, c_fp]
    L.nalo_ba_get_idepth_zero.argtypes = [vp, c_fp]
    L.nalo_ba_calc_l_energy.argtypes = [vp, c_dp]
    L.nalo_ba_plane_scale_fix.argtypes = [vp, C.c_double, c_dp, c_dp]
    L.nalo_ba_sw_gray_optimize.argtypes = [vp, c_dp, c_ip]
    L.nalo_ba_calc_m_energy.argtypes = [vp, c_dp]
    L.nalo_ba_optimize_stats.argtypes = [vp, c_ip, c_ip]
    L.nalo_get_settings.argtypes = [vp, C.POINTER(Settings)]
    L.nalo_set_settings.argtypes = [vp, C.POINTER(Settings)]
    L.nalo_init_set_first.argtypes = [vp, C.c_int, c_ip, c_ip]
    L.nalo_init_track_frame.argtypes = [vp, C.c_int, C.c_float, C.c_float, c_ip]
    L.nalo_init_get_state.argtypes = [vp, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip]
    L.nalo_init_get_points.argtypes = [vp, C.c_int, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp, c_u8p, c_fp, c_fp, c_fp, c_fp, c_ip, c_fp, c_ip, c_fp]
    L.nalo_trk_last_evals.argtypes = [vp, c_ip, c_ip]
    L.nalo_init_set_state.argtypes = [vp, c_dp, c_dp, C.c_int, C.c_int, C.c_int]
    L.nalo_init_set_points.argtypes = [vp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_u8p, c_fp, c_fp, c_fp, c_fp, c_fp, c_u8p, c_fp]
    L.nalo_init_get_carried.argtypes = [vp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_u8p]
    L.nalo_init_sweep.argtypes = [vp, C.c_int, C.c_int]
    L.nalo_pixsel_make_hists.argtypes = [vp, C.c_int, c_fp, c_fp]
    L.nalo_pixsel_set_random.argtypes = [vp, c_u8p, c_ip]
    L.nalo_pixsel_select.argtypes = [vp, C.c_int, C.c_int, C.c_float, c_fp, c_ip]
    L.nalo_pixsel_make_maps.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_float, c_ip, c_fp, c_ip]
    L.nalo_pixsel_make_maps_lidar.argtypes = [vp, C.c_int, C.c_float, C.c_int, c_fp, c_ip]
    L.nalo_pixsel_get_selected.argtypes = [vp, C.c_int, c_ip, c_u8p, c_ip]
    L.nalo_dist_make_map.argtypes = [vp, C.c_int, c_fp, c_fp, c_fp]
    L.nalo_imm_create.argtypes = [vp, C.c_int, C.c_int, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp]
    L.nalo_imm_trace.argtypes = [vp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip, c_fp, c_fp, c_fp]
    L.nalo_imm_resident_set.argtypes = [vp, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_ip, c_fp, c_fp, c_ip, c_fp]
    L.nalo_imm_resident_trace.argtypes = [vp, C.c_int, C.c_int, c_fp, c_fp, c_fp]
    L.nalo_imm_resident_get.argtypes = [vp, c_fp, c_fp, c_ip, c_fp, c_fp, c_fp]
    L.nalo_imm_optimize.argtypes = [vp, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int, c_ip, c_fp, c_u8p]
    L.nalo_imm_resident_optimize.argtypes = [vp, C.c_int, c_ip, C.c_int, c_ip, c_fp, c_u8p]
    L.nalo_ba_restore.argtypes = [vp]
    L.nalo_dense_make_map.argtypes = [vp, C.c_int, c_fp, C.c_float, c_dp, C.c_int, c_ip, c_ip, c_ip, c_fp, c_fp, c_u8p, c_ip, c_ip]
    L.nalo_profile_enable.argtypes = [vp, C.c_int]
    L.nalo_profile_reset.argtypes = [vp]
    L.nalo_profile_select.argtypes = [vp, C.c_char_p]
    L.nalo_profile_get.argtypes = [vp, C.c_char_p, c_dp, c_ip]
    L.nalo_profile_samples.argtypes = [vp, C.c_char_p, c_fp, C.c_int, c_ip]
    L.nalo_profile_sample.argtypes = [vp, C.c_int]
    L.nalo_hbm_calibrate.argtypes = [vp, C.c_size_t, C.c_int, c_dp, c_dp]
    _LIB = L
    return L


def _f(a):
    return None if a is None else a.ctypes.data_as(c_fp)


def _d(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(c_ip)


def _u8(a):
    return None if a is None else a.ctypes.data_as(c_u8p)


class NaloError(RuntimeError):
    pass


def shard_points(host, u, v, W, img_w, img_h, rank, world):
    """indices (ascending) of the active points rank `rank` of `world` keeps: nalo_shard_points, host code (no device needed)"""
    L = load()
    h = np.ascontiguousarray(host, np.int32)
    uu, vv = np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32)
    keep = np.zeros(len(h), np.int32)
    n = L.nalo_shard_points(len(h), int(W), _i(h), _f(uu), _f(vv), int(img_w), int(img_h), int(rank), int(world), _i(keep))
    if n < 0:
        raise NaloError("nalo_shard_points: bad argument (%d)" % n)
    return keep[:n].copy()


def rccl_unique_id():
    """128-byte ncclUniqueId (rank 0 draws it, the caller ships it to the other ranks)"""
    buf = C.create_string_buffer(128)
    rc = load().nalo_rccl_unique_id(buf)
    if rc != 0:
        raise NaloError("nalo_rccl_unique_id failed (%d): librccl missing?" % rc)
    return buf.raw


class Context:
    """Thin object wrapper over nalo_ctx*."""

    def __init__(self, w, h, K, n_slots, levels=0, device=0):
        self.L = load()
        self.h_ = C.c_void_p()
        Kf = np.asarray(K, np.float32)
        rc = self.L.nalo_create(C.byref(self.h_), device, w, h, levels, _f(Kf), n_slots)
        if rc != 0:
            raise NaloError("nalo_create failed (%d): needs a gfx950 device, no CPU fallback" % rc)
        self.w, self.h, self.K = w, h, tuple(float(k) for k in K)
        self.levels = self.L.nalo_levels(self.h_)
        self.W = 0
        self.P = 0
        self._hook = None
        self._pinned = []

    def close(self):
        if self.h_:
            self.L.nalo_destroy(self.h_)
            self.h_ = C.c_void_p()
            for p in self._pinned:
                self.L.nalo_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise NaloError("nalo error %d: %s" % (rc, self.L.nalo_last_error(self.h_).decode()))

    def sync(self):
        self._ck(self.L.nalo_sync(self.h_))

    # ---- frames
    def frame_upload(self, slot, img, mask=None, bgr=None, gammaB=None):
        img = np.ascontiguousarray(img, np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.float32)
        b = None if bgr is None else np.ascontiguousarray(bgr, np.uint8)
        g = None if gammaB is None else np.ascontiguousarray(gammaB, np.float32)
        assert g is None or g.size == 256
        self._ck(self.L.nalo_frame_upload(self.h_, slot, _f(img), _f(m), _u8(b), _f(g)))

    def frame_upload_async(self, slot, img, mask=None, bgr=None, gammaB=None):
        """img (and mask / bgr) must be contiguous arrays of the right dtype that stay alive and untouched until frame_wait(slot): no conversion, no copy
        here (pinned_array() gives page-locked arrays, the only kind the copy engine reads asynchronously)"""
        assert img.dtype == np.float32 and img.flags.c_contiguous and img.size == self.w * self.h
        assert mask is None or (mask.dtype == np.float32 and mask.flags.c_contiguous)
        assert bgr is None or (bgr.dtype == np.uint8 and bgr.flags.c_contiguous)
        assert gammaB is None or (gammaB.dtype == np.float32 and gammaB.size == 256)
        self._ck(self.L.nalo_frame_upload_async(self.h_, slot, _f(img), _f(mask), _u8(bgr), _f(gammaB)))

    def frame_wait(self, slot):
        self._ck(self.L.nalo_frame_wait(self.h_, slot))

    def pinned_array(self, shape, dtype=np.float32):
        """numpy array over page-locked host memory from nalo_host_alloc (freed when the context closes)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.L.nalo_host_alloc(n)
        if not p:
            raise NaloError("nalo_host_alloc(%d) failed" % n)
        self._pinned.append(p)
        buf = (C.c_char * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def frame_rebuild(self, slot):
        self._ck(self.L.nalo_frame_rebuild(self.h_, slot))

    def frame_download(self, slot, lvl):
        n = (self.w >> lvl) * (self.h >> lvl)
        dI, ab = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        self._ck(self.L.nalo_frame_download(self.h_, slot, lvl, _f(dI), _f(ab)))
        return dI, ab

    # ---- tracker
    def trk_set_ref(self, slot, Ku, Kv, new_idepth, HdiF):
        a = [np.ascontiguousarray(x, np.float32) for x in (Ku, Kv, new_idepth, HdiF)]
        self._ck(self.L.nalo_trk_set_ref(self.h_, slot, len(a[0]), *[_f(x) for x in a]))

    def trk_ref_upload(self, Ku, Kv, new_idepth, HdiF):
        """the inputs of setCoarseTrackingRef resident on the device (nalo_trk_ref_upload); trk_set_ref_resident(slot) then builds the reference from them"""
        a = [np.ascontiguousarray(x, np.float32) for x in (Ku, Kv, new_idepth, HdiF)]
        self._ck(self.L.nalo_trk_ref_upload(self.h_, len(a[0]), *[_f(x) for x in a]))

    def trk_set_ref_resident(self, slot):
        self._ck(self.L.nalo_trk_set_ref_resident(self.h_, slot))

    def trk_set_pc(self, slot, lvl, u, v, idepth, color):
        a = [np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, color)]
        self._ck(self.L.nalo_trk_set_pc(self.h_, slot, lvl, len(a[0]), *[_f(x) for x in a]))

    def trk_get_pc(self, lvl):
        n = C.c_int(0)
        self._ck(self.L.nalo_trk_get_pc(self.h_, lvl, C.byref(n), None, None, None, None))
        out = [np.zeros(n.value, np.float32) for _ in range(4)]
        self._ck(self.L.nalo_trk_get_pc(self.h_, lvl, C.byref(n), *[_f(o) for o in out]))
        return out

    def trk_get_depth(self, lvl):
        n = (self.w >> lvl) * (self.h >> lvl)
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self._ck(self.L.nalo_trk_get_depth(self.h_, lvl, _f(a), _f(b)))
        return a, b

    def trk_eval(self, slot_new, lvl, T, affLL, b0, cutoff, want_gs=True):
        T = np.ascontiguousarray(T, np.float64).reshape(3, 4)
        R = np.ascontiguousarray(T[:, :3]).reshape(-1)
        t = np.ascontiguousarray(T[:, 3])
        st, H, b = np.zeros(6), np.zeros(64), np.zeros(8)
        self._ck(self.L.nalo_trk_eval(self.h_, slot_new, lvl, _d(R), _d(t), _f(np.asarray(affLL, np.float32)), b0, cutoff,
                                      int(want_gs), _d(st), _d(H), _d(b)))
        return st, H.reshape(8, 8), b

    def trk_set_shard(self, rank, world, fn=None, stream_ordered=False):
        """sharded tracker: this context evaluates rank / world of every level's points; fn(device_ptr:int, n:int) sums n doubles in place across ranks. world = 1: off"""
        self._trk_hook = ALLREDUCE_FN(lambda user, ptr, n: fn(ptr, n)) if fn is not None else C.cast(None, ALLREDUCE_FN)
        self.L.nalo_trk_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p, C.c_int]
        self._ck(self.L.nalo_trk_set_shard(self.h_, int(rank), int(world), self._trk_hook, None, int(bool(stream_ordered))))

    def trk_track(self, slot_new, T0, aff0, ref_aff, exposures, coarsest, min_res=None):
        T = np.ascontiguousarray(T0, np.float64).reshape(-1).copy()
        aff = np.array(aff0, np.float64)
        mr = np.full(5, np.nan) if min_res is None else np.asarray(min_res, np.float64)
        lr, lf = np.zeros(5), np.zeros(3)
        ok, ne = C.c_int(0), C.c_int(0)
        self._ck(self.L.nalo_trk_track(self.h_, slot_new, _d(T), _d(aff), _d(np.asarray(ref_aff, np.float64)),
                                       _f(np.asarray(exposures, np.float32)), coarsest, _d(mr), _d(lr), _d(lf), C.byref(ok), C.byref(ne)))
        return ok.value, T.reshape(3, 4), aff, lr, lf, ne.value

    # ---- BA
    def ba_set_window(self, slots, evalPT, aff=None, exposure=None, th=None, frame_ids=None, state6=None, calib=None, calib_zero=None,
                      states=None, states_zero=None):
        """states / states_zero ([W][10], unscaled FrameHessian::state / state_zero) give the general form a running window needs; without them the
        frames are set like FrameHessian::setEvalPT_scaled (state_zero = [0.., a/SCALE_A, b/SCALE_B, 0, 0]) plus an optional state6."""
        W = len(slots)
        arr = (FrameState * W)()
        for i in range(W):
            fs = arr[i]
            fs.slot = int(slots[i])
            fs.frame_id = int(i if frame_ids is None else frame_ids[i])
            e = np.ascontiguousarray(evalPT[i], np.float64).reshape(-1)
            for k in range(12):
                fs.worldToCam_evalPT[k] = e[k]
            if states is not None:
                for k in range(10):
                    fs.state[k] = float(states[i][k])
                    fs.state_zero[k] = float(states_zero[i][k])
            else:
                a, b = (0.0, 0.0) if aff is None else aff[i]
                # FrameHessian::setEvalPT_scaled (HessianBlocks.h:247-255): state = [0.., a/SCALE_A, b/SCALE_B, 0, 0], state_zero = state
                st = np.zeros(10)
                st[6] = np.float32(1.0 / 10.0) * a
                st[7] = np.float32(1.0 / 1000.0) * b
                for k in range(10):
                    fs.state_zero[k] = st[k]
                if state6 is not None:
                    st[:6] = state6[i]
                for k in range(10):
                    fs.state[k] = st[k]
            fs.ab_exposure = 1.0 if exposure is None else float(exposure[i])
            fs.frameEnergyTH = 8 * 8 * 8.0 if th is None else float(th[i])
        cal = np.asarray(self.K if calib is None else calib, np.float64)
        calz = cal if calib_zero is None else np.asarray(calib_zero, np.float64)
        self._ck(self.L.nalo_ba_set_window(self.h_, W, arr, _d(cal), _d(calz)))
        self.W = W

    def ba_set_points(self, host, u, v, idepth, color, weights, has_prior=None, idepth_zero=None):
        a = [np.ascontiguousarray(host, np.int32)] + [np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, color, weights)]
        hp = None if has_prior is None else np.ascontiguousarray(has_prior, np.int32)
        iz = None if idepth_zero is None else np.ascontiguousarray(idepth_zero, np.float32)
        self.P = len(a[0])
        self._ck(self.L.nalo_ba_set_points(self.h_, self.P, _i(a[0]), _f(a[1]), _f(a[2]), _f(a[3]), _f(iz), _f(a[4]), _f(a[5]), _i(hp)))

    def ba_set_prior(self, HM=None, bM=None):
        H = None if HM is None else np.ascontiguousarray(HM, np.float64)
        b = None if bM is None else np.ascontiguousarray(bM, np.float64)
        assert H is None or H.size == self.n * self.n
        self._ck(self.L.nalo_ba_set_prior(self.h_, _d(H), _d(b)))

    def ba_get_prior(self):
        n = self.n
        H, b = np.zeros(n * n), np.zeros(n)
        self._ck(self.L.nalo_ba_get_prior(self.h_, _d(H), _d(b)))
        return H.reshape(n, n), b

    def ba_marginalize_frame(self, idx):
        """EnergyFunctional::marginalizeFrame on HM/bM; the window shrinks by one frame (re-issue set_window / set_points / set_residuals next)"""
        self._ck(self.L.nalo_ba_marginalize_frame(self.h_, int(idx)))
        self.W -= 1

    def ba_set_prior_carry(self, on):
        """declare the context one continuing EnergyFunctional: every ba_set_window keeps (same frames) or extends (one frame appended) HM / bM"""
        self._ck(self.L.nalo_ba_set_prior_carry(self.h_, int(bool(on))))

    def ba_set_residuals(self, exists):
        self._ck(self.L.nalo_ba_set_residuals(self.h_, _u8(np.ascontiguousarray(exists, np.uint8))))

    @property
    def n(self):
        return 8 * self.W + 4

    def ba_linearize(self, fix=False):
        e = C.c_double(0)
        self._ck(self.L.nalo_ba_linearize(self.h_, int(fix), C.byref(e)))
        return e.value

    def ba_accumulate(self, mode):
        H, b = np.zeros(self.n * self.n), np.zeros(self.n)
        self._ck(self.L.nalo_ba_accumulate(self.h_, mode, _d(H), _d(b)))
        return H.reshape(self.n, self.n), b

    def ba_accumulate_sc(self, shift=True):
        H, b = np.zeros(self.n * self.n), np.zeros(self.n)
        self._ck(self.L.nalo_ba_accumulate_sc(self.h_, int(shift), _d(H), _d(b)))
        return H.reshape(self.n, self.n), b

    def ba_solve_system(self, iteration, lam=1e-5):
        x = np.zeros(self.n)
        self._ck(self.L.nalo_ba_solve_system(self.h_, iteration, lam, _d(x)))
        return x

    def ba_backup_state(self):
        self._ck(self.L.nalo_ba_backup_state(self.h_))

    def ba_do_step(self, f=1.0):
        cb = C.c_int(0)
        self._ck(self.L.nalo_ba_do_step(self.h_, f, f, f, f, f, C.byref(cb)))
        return cb.value

    def ba_optimize(self, its=6, never_break=False):
        r = C.c_double(0)
        self._ck(self.L.nalo_ba_optimize(self.h_, its, int(never_break), C.byref(r)))
        return r.value

    def ba_marginalize_points(self, flags):
        n = self.n
        M, Mb, Ms, Mbs = np.zeros(n * n), np.zeros(n), np.zeros(n * n), np.zeros(n)
        self._ck(self.L.nalo_ba_marginalize_points(self.h_, _u8(np.ascontiguousarray(flags, np.uint8)), _d(M), _d(Mb), _d(Ms), _d(Mbs)))
        return M.reshape(n, n), Mb, Ms.reshape(n, n), Mbs

    def ba_get_frames(self):
        arr = (FrameState * self.W)()
        w2c = np.zeros((self.W, 12))
        cal = np.zeros(4)
        self._ck(self.L.nalo_ba_get_frames(self.h_, arr, _d(w2c), _d(cal)))
        return arr, w2c.reshape(self.W, 3, 4), cal

    def ba_get_points(self):
        P = self.P
        o = {k: np.zeros(P, np.float32) for k in ("idepth", "step", "HdiF", "bdSumF", "Hdd", "bd")}
        o["Hcd"] = np.zeros((P, 4), np.float32)
        o["maxRelBaseline"] = np.zeros(P, np.float32)
        self._ck(self.L.nalo_ba_get_points(self.h_, _f(o["idepth"]), _f(o["step"]), _f(o["HdiF"]), _f(o["bdSumF"]), _f(o["Hdd"]),
                                           _f(o["bd"]), _f(o["Hcd"]), _f(o["maxRelBaseline"])))
        return o

    def ba_get_residuals(self):
        n = self.P * self.W
        st, ac = np.zeros(n, np.int8), np.zeros(n, np.uint8)
        jp, en, cp = np.zeros((n, 8), np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
        self._ck(self.L.nalo_ba_get_residuals(self.h_, st.ctypes.data_as(c_i8p), _u8(ac), _f(jp), _f(en), _f(cp)))
        s = (self.P, self.W)
        return st.reshape(s), ac.reshape(s), jp.reshape(s + (8,)), en.reshape(s), cp.reshape(s + (3,))

    def ba_get_acc13(self):
        a = np.zeros(self.W * self.W * 169)
        self._ck(self.L.nalo_ba_get_acc13(self.h_, _d(a)))
        return a.reshape(self.W * self.W, 13, 13)

    def ba_counts(self):
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(self.L.nalo_ba_counts(self.h_, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- two-frame initialiser
    def init_calc_res_and_gs(self, slot_first, slot_new, lvl, refToNew, aff, pts, alphaW=150.0 * 150.0, alphaK=2.5 * 2.5, couplingWeight=1.0):
        f = lambda a: np.ascontiguousarray(a, np.float32)
        n = len(pts["u"])
        ig = np.ascontiguousarray(pts["isGood"], np.uint8)
        ign, en, ms = np.zeros(n, np.uint8), np.zeros((n, 2), np.float32), np.zeros(n, np.float32)
        lh, jb = f(pts["lastHessian_new"]).copy(), f(pts["Jb"]).copy()
        H, b, Hs, bs, E3 = np.zeros(64), np.zeros(8), np.zeros(64), np.zeros(8), np.zeros(3)
        a = [f(pts[k]) for k in ("u", "v", "idepth_new", "iR")]
        eng, oth = f(pts["energy"]), f(pts["outlierTH"])
        T = np.ascontiguousarray(refToNew, np.float64).reshape(-1)
        af = np.ascontiguousarray(aff, np.float64)
        self._ck(self.L.nalo_init_calc_res_and_gs(self.h_, slot_first, slot_new, lvl, n, *[_f(x) for x in a], _u8(ig), _f(eng), _f(oth), _d(T), _d(af),
                                                  alphaW, alphaK, couplingWeight, _u8(ign), _f(en), _f(ms), _f(lh), _f(jb), _d(H), _d(b), _d(Hs), _d(bs), _d(E3)))
        return dict(H=H.reshape(8, 8), b=b, Hsc=Hs.reshape(8, 8), bsc=bs, E3=E3, isGood_new=ign, energy_new=en, maxstep=ms, lastHessian_new=lh, Jb=jb)

    def init_do_step(self, isGood, Jb, maxstep, idepth, lam, inc, idepth_new):
        f = lambda a: np.ascontiguousarray(a, np.float32)
        out = f(idepth_new).copy()
        self._ck(self.L.nalo_init_do_step(self.h_, len(out), _u8(np.ascontiguousarray(isGood, np.uint8)), _f(f(Jb)), _f(f(maxstep)), _f(f(idepth)), float(lam), _f(f(inc)), _f(out)))
        return out

    def dense_make_map(self, slot, plane, mask_value, camToWorld, cap=100000):
        """DenseMapping::updateMap bbox scan + makeMap -> dict(n, accept, rect, u, v, idepth, color, bgr)"""
        rect = np.zeros(4, np.int32)
        u, v = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        idp, col, bgr = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
        n, acc = np.zeros(1, np.int32), np.zeros(1, np.int32)
        self._ck(self.L.nalo_dense_make_map(self.h_, slot, _f(np.ascontiguousarray(plane, np.float32)), C.c_float(mask_value), _d(np.ascontiguousarray(camToWorld, np.float64).reshape(-1)),
                                            cap, _i(rect), _i(u), _i(v), _f(idp), _f(col), _u8(bgr), _i(n), _i(acc)))
        k = min(int(n[0]), cap)
        return dict(n=int(n[0]), accept=int(acc[0]), rect=rect, u=u[:k], v=v[:k], idepth=idp[:k], color=col[:k], bgr=bgr[:k])

    def trk_append_plane_points(self, dirv, dis, ref_color, rect):
        n = np.zeros(1, np.int32)
        self._ck(self.L.nalo_trk_append_plane_points(self.h_, _f(np.ascontiguousarray(dirv, np.float32)), C.c_float(dis), int(ref_color), _i(np.ascontiguousarray(rect, np.int32)), _i(n)))
        return int(n[0])

    def undist_set(self, wOrg, hOrg, G=None, vinv=None, photometric=0, remapX=None, remapY=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        G, vinv, remapX, remapY = f(G), f(vinv), f(remapX), f(remapY)
        self._ck(self.L.nalo_undist_set(self.h_, wOrg, hOrg, None if G is None else _f(G), 0 if G is None else G.size, None if vinv is None else _f(vinv), int(photometric),
                                        None if remapX is None else _f(remapX), None if remapY is None else _f(remapY)))

    def frame_upload_raw(self, slot, raw, exposure=1.0, factor=1.0, mask_org=None, bgr_org=None, gammaB=None):
        raw = np.ascontiguousarray(raw)
        assert raw.dtype in (np.uint8, np.uint16)
        m = None if mask_org is None else np.ascontiguousarray(mask_org, np.uint8)
        b = None if bgr_org is None else np.ascontiguousarray(bgr_org, np.uint8)
        g = None if gammaB is None else np.ascontiguousarray(gammaB, np.float32)
        self._ck(self.L.nalo_frame_upload_raw(self.h_, slot, raw.ctypes.data_as(C.c_void_p), raw.dtype.itemsize, C.c_float(exposure), C.c_float(factor),
                                              None if m is None else _u8(m), None if b is None else _u8(b), None if g is None else _f(g)))

    def frame_upload_raw_async(self, slot, raw, exposure=1.0, factor=1.0):
        """raw: a contiguous uint8 / uint16 array that stays alive and untouched until frame_wait(slot) (pinned: pinned_array(..., dtype))"""
        assert raw.flags.c_contiguous and raw.dtype in (np.uint8, np.uint16)
        self._ck(self.L.nalo_frame_upload_raw_async(self.h_, slot, raw.ctypes.data_as(C.c_void_p), raw.dtype.itemsize, C.c_float(exposure), C.c_float(factor), None))

    def get_settings(self):
        st = Settings()
        self._ck(self.L.nalo_get_settings(self.h_, C.byref(st)))
        return {k: getattr(st, k) for k, _ in Settings._fields_}

    def set_settings(self, force_accept_step=None, affine_opt_mode_a=None, affine_opt_mode_b=None, min_opt_iterations=None):
        st = Settings()
        self._ck(self.L.nalo_get_settings(self.h_, C.byref(st)))
        if force_accept_step is not None: st.forceAcceptStep = int(force_accept_step)
        if affine_opt_mode_a is not None: st.affineOptModeA = float(affine_opt_mode_a)
        if affine_opt_mode_b is not None: st.affineOptModeB = float(affine_opt_mode_b)
        if min_opt_iterations is not None: st.minOptIterations = int(min_opt_iterations)
        self._ck(self.L.nalo_set_settings(self.h_, C.byref(st)))
        return st

    def ba_plane_scale_fix(self, localscale, camToTrackingRef, trackingRef_camToWorld):
        a = np.ascontiguousarray(camToTrackingRef, np.float64).reshape(-1); b = np.ascontiguousarray(trackingRef_camToWorld, np.float64).reshape(-1)
        self._ck(self.L.nalo_ba_plane_scale_fix(self.h_, float(localscale), _d(a), _d(b)))

    def ba_sw_gray_optimize(self):
        cost, n = np.zeros(1), np.zeros(1, np.int32)
        self._ck(self.L.nalo_ba_sw_gray_optimize(self.h_, _d(cost), _i(n)))
        return float(cost[0]), int(n[0])

    def ba_get_idepth_zero(self, P):
        o = np.zeros(P, np.float32)
        self._ck(self.L.nalo_ba_get_idepth_zero(self.h_, _f(o)))
        return o

    def ba_calc_l_energy(self):
        e = np.zeros(1)
        self._ck(self.L.nalo_ba_calc_l_energy(self.h_, _d(e)))
        return float(e[0])

    def ba_calc_m_energy(self):
        e = np.zeros(1)
        self._ck(self.L.nalo_ba_calc_m_energy(self.h_, _d(e)))
        return float(e[0])

    def ba_optimize_stats(self):
        a = np.zeros(2, np.int32)
        self._ck(self.L.nalo_ba_optimize_stats(self.h_, a[0:1].ctypes.data_as(C.POINTER(C.c_int)), a[1:2].ctypes.data_as(C.POINTER(C.c_int))))
        return int(a[0]), int(a[1])

    def init_set_first(self, slot_first, sparsityFactor=5):
        """CoarseInitializer::setFirst -> (points per level, the updated global sparsityFactor)"""
        sf, num = np.array([sparsityFactor], np.int32), np.zeros(6, np.int32)
        self._ck(self.L.nalo_init_set_first(self.h_, slot_first, _i(sf), _i(num)))
        return num[:self.levels].copy(), int(sf[0])

    def init_track_frame(self, slot_new, exposure_first=1.0, exposure_new=1.0):
        ok = np.zeros(1, np.int32)
        self._ck(self.L.nalo_init_track_frame(self.h_, slot_new, C.c_float(exposure_first), C.c_float(exposure_new), _i(ok)))
        return bool(ok[0])

    def init_state(self):
        T, aff, st = np.zeros(12), np.zeros(2), np.zeros(4, np.int32)
        p = lambda k: st[k:k + 1].ctypes.data_as(C.POINTER(C.c_int))
        self._ck(self.L.nalo_init_get_state(self.h_, _d(T), _d(aff), p(0), p(1), p(2), p(3)))
        return dict(thisToNext=T.reshape(3, 4), aff=aff, snapped=bool(st[0]), frameID=int(st[1]), snappedAt=int(st[2]), n_evals=int(st[3]))

    def init_set_carried(self, car):
        """car = orc.Initializer.carried(): the pose / affine / snap state and, per level, orc.Initializer.CARRIED"""
        s = car["state"]
        self._ck(self.L.nalo_init_set_state(self.h_, _d(np.ascontiguousarray(s["thisToNext"], np.float64).reshape(-1)), _d(np.ascontiguousarray(s["aff"], np.float64)),
                                            int(s["snapped"]), int(s["frameID"]), int(s["snappedAt"])))
        for l, p in enumerate(car["points"]):
            a = {k: np.ascontiguousarray(p[k], np.uint8 if k.startswith("isGood") else np.float32) for k in p}
            self._ck(self.L.nalo_init_set_points(self.h_, l, len(a["idepth"]), _f(a["idepth"]), _f(a["idepth_new"]), _f(a["iR"]), _u8(a["isGood"]), _f(a["lastHessian"]),
                                                 _f(a["energy"]), _f(a["maxstep"]), _f(a["lastHessian_new"]), _f(a["energy_new"]), _u8(a["isGood_new"]), _f(a["iRSumNum"])))

    SWEEPS = dict(opt_reg=0, propagate_up=1, propagate_down=2, reset_points=3)

    def init_sweep(self, which, lvl):
        """one of trackFrame's sweeps alone: optReg(lvl), propagateUp(srcLvl), propagateDown(srcLvl), resetPoints(lvl)"""
        self._ck(self.L.nalo_init_sweep(self.h_, self.SWEEPS[which], lvl))

    def init_carried(self, lvl):
        """the Pnt members nalo_init_set_points writes and init_points() does not return"""
        n = len(self.init_points(lvl)["u"]); m = max(n, 1)
        o = dict(idepth_new=np.zeros(m, np.float32), maxstep=np.zeros(m, np.float32), lastHessian_new=np.zeros(m, np.float32), energy_new=np.zeros((m, 2), np.float32),
                 isGood_new=np.zeros(m, np.uint8))
        self._ck(self.L.nalo_init_get_carried(self.h_, lvl, n, _f(o["idepth_new"]), _f(o["maxstep"]), _f(o["lastHessian_new"]), _f(o["energy_new"]), _u8(o["isGood_new"])))
        return {k: v[:n] for k, v in o.items()}

    def init_points(self, lvl):
        n = np.zeros(1, np.int32)
        self._ck(self.L.nalo_init_get_points(self.h_, lvl, 0, _i(n), *([None] * 13)))
        n = int(n[0]); m = max(n, 1)
        f = lambda k=1: np.zeros((m, k) if k > 1 else m, np.float32)
        o = dict(u=f(), v=f(), idepth=f(), iR=f(), isGood=np.zeros(m, np.uint8), lastHessian=f(), energy=f(2), my_type=f(), outlierTH=f(), parent=np.zeros(m, np.int32),
                 parentDist=f(), neighbours=np.zeros((m, 10), np.int32), neighboursDist=f(10))
        nn = np.zeros(1, np.int32)
        self._ck(self.L.nalo_init_get_points(self.h_, lvl, n, _i(nn), _f(o["u"]), _f(o["v"]), _f(o["idepth"]), _f(o["iR"]), _u8(o["isGood"]), _f(o["lastHessian"]), _f(o["energy"]),
                                             _f(o["my_type"]), _f(o["outlierTH"]), _i(o["parent"]), _f(o["parentDist"]), _i(o["neighbours"]), _f(o["neighboursDist"])))
        return {k: a[:n] for k, a in o.items()}

    def pixsel_make_hists(self, slot):
        nb = (self.w // 32) * (self.h // 32)
        ths, sm = np.zeros(nb, np.float32), np.zeros(nb, np.float32)
        self._ck(self.L.nalo_pixsel_make_hists(self.h_, slot, _f(ths), _f(sm)))
        return ths, sm

    def pixsel_set_random(self, randomPattern, mask_draws=None):
        rp = np.ascontiguousarray(randomPattern, np.uint8)
        assert rp.size == self.w * self.h
        dr = None if mask_draws is None else np.ascontiguousarray(mask_draws, np.int32)
        self._ck(self.L.nalo_pixsel_set_random(self.h_, _u8(rp), None if dr is None else _i(dr)))

    def pixsel_select(self, slot, pot, thFactor=1.0):
        m, n = np.zeros((self.h, self.w), np.float32), np.zeros(3, np.int32)
        self._ck(self.L.nalo_pixsel_select(self.h_, slot, pot, float(thFactor), _f(m), _i(n)))
        return m, n

    def pixsel_make_maps(self, slot, density, potential, recursionsLeft=1, thFactor=1.0):
        """-> (map [h,w], numHaveSub, new currentPotential)"""
        m, pot, num = np.zeros((self.h, self.w), np.float32), np.array([potential], np.int32), np.zeros(1, np.int32)
        self._ck(self.L.nalo_pixsel_make_maps(self.h_, slot, float(density), recursionsLeft, float(thFactor), _i(pot), _f(m), _i(num)))
        return m, int(num[0]), int(pot[0])

    def pixsel_make_maps_lidar(self, slot, potential, thFactor=1.0):
        m, num = np.zeros((self.h, self.w), np.float32), np.zeros(1, np.int32)
        self._ck(self.L.nalo_pixsel_make_maps_lidar(self.h_, slot, float(thFactor), potential, _f(m), _i(num)))
        return m, int(num[0])

    def pixsel_get_selected(self):
        n = np.zeros(1, np.int32)
        self._ck(self.L.nalo_pixsel_get_selected(self.h_, 0, None, None, _i(n)))
        idx, st = np.zeros(max(int(n[0]), 1), np.int32), np.zeros(max(int(n[0]), 1), np.uint8)
        self._ck(self.L.nalo_pixsel_get_selected(self.h_, int(n[0]), _i(idx), _u8(st), _i(n)))
        return idx[:n[0]], st[:n[0]]

    def dist_make_map(self, frame, KRKi, Kt):
        out = np.zeros((self.h >> 1, self.w >> 1), np.float32)
        self._ck(self.L.nalo_dist_make_map(self.h_, frame, _f(np.ascontiguousarray(KRKi, np.float32)), _f(np.ascontiguousarray(Kt, np.float32)), _f(out)))
        return out

    # ---- immature points
    def imm_create(self, slot_host, u, v):
        n = len(u)
        ui, vi = np.ascontiguousarray(u, np.int32), np.ascontiguousarray(v, np.int32)
        color, weights, gradH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        self._ck(self.L.nalo_imm_create(self.h_, slot_host, n, _i(ui), _i(vi), _f(color), _f(weights), _f(gradH), _f(eth)))
        return color, weights, gradH, eth

    def imm_trace(self, slot_new, u, v, color, weights, gradH, energyTH, host_idx, KRKi, Kt, aff, idmin, idmax, status, quality):
        n = len(u)
        f = lambda a: np.ascontiguousarray(a, np.float32).copy()
        idmin, idmax, quality = f(idmin), f(idmax), f(quality)
        status = np.ascontiguousarray(status, np.int32).copy()
        uv, li = np.zeros((n, 2), np.float32), np.zeros(n, np.float32)
        hi = np.ascontiguousarray(host_idx, np.int32)
        a = [f(x) for x in (u, v, color, weights, gradH, energyTH)]
        k = [f(x) for x in (KRKi, Kt, aff)]
        self._ck(self.L.nalo_imm_trace(self.h_, slot_new, n, *[_f(x) for x in a], _i(hi), len(k[0].reshape(-1, 9)), *[_f(x) for x in k],
                                       _f(idmin), _f(idmax), _i(status), _f(quality), _f(uv), _f(li)))
        return idmin, idmax, status, quality, uv, li

    def imm_resident_set(self, u, v, color, weights, gradH, energyTH, host_idx, idmin, idmax, status, quality):
        f = lambda a: np.ascontiguousarray(a, np.float32)
        a = [f(x) for x in (u, v, color, weights, gradH, energyTH)]
        self._imm_n = len(a[0])
        self._ck(self.L.nalo_imm_resident_set(self.h_, self._imm_n, *[_f(x) for x in a], _i(np.ascontiguousarray(host_idx, np.int32)), _f(f(idmin)), _f(f(idmax)),
                                              _i(np.ascontiguousarray(status, np.int32)), _f(f(quality))))

    def imm_resident_trace(self, slot_new, KRKi, Kt, aff):
        k = [np.ascontiguousarray(x, np.float32) for x in (KRKi, Kt, aff)]
        self._ck(self.L.nalo_imm_resident_trace(self.h_, slot_new, len(k[0].reshape(-1, 9)), *[_f(x) for x in k]))

    def imm_resident_get(self):
        n = self._imm_n
        idmin, idmax, quality, li = [np.zeros(n, np.float32) for _ in range(4)]
        status, uv = np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
        self._ck(self.L.nalo_imm_resident_get(self.h_, _f(idmin), _f(idmax), _i(status), _f(quality), _f(uv), _f(li)))
        return idmin, idmax, status, quality, uv, li

    def imm_optimize(self, host, u, v, color, weights, energyTH, idmin, idmax, min_obs):
        n = len(u)
        f = lambda a: np.ascontiguousarray(a, np.float32)
        a = [f(x) for x in (u, v, color, weights, energyTH, idmin, idmax)]
        hi = np.ascontiguousarray(host, np.int32)
        res, idp, rin = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, self.W), np.uint8)
        self._ck(self.L.nalo_imm_optimize(self.h_, n, _i(hi), *[_f(x) for x in a], int(min_obs), _i(res), _f(idp), _u8(rin)))
        return res, idp, rin

    def imm_resident_optimize(self, sel, min_obs, n_all=None):
        """optimizeImmaturePoint for points of the device-resident set: sel = their indices (None: all n_all of them)"""
        if sel is None:
            n, sp = int(n_all), None
        else:
            si = np.ascontiguousarray(sel, np.int32); n, sp = len(si), _i(si)
        res, idp, rin = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, self.W), np.uint8)
        self._ck(self.L.nalo_imm_resident_optimize(self.h_, n, sp, int(min_obs), _i(res), _f(idp), _u8(rin)))
        return res, idp, rin

    def ba_snapshot(self):
        self._ck(self.L.nalo_ba_snapshot(self.h_))

    def ba_restore(self):
        self._ck(self.L.nalo_ba_restore(self.h_))

    @property
    def stream(self):
        """hipStream_t (as int) every kernel of this context is launched on"""
        return int(self.L.nalo_stream(self.h_) or 0)

    @property
    def side_stream(self):
        """hipStream_t (as int) of the context's second stream (the threshold's histogram sums of a sharded window run there)"""
        return int(self.L.nalo_side_stream(self.h_) or 0)

    def ba_set_allreduce(self, fn, stream_ordered=False, fn_side=None):
        """fn(device_ptr:int, n:int) sums n doubles in place across ranks. stream_ordered: fn enqueues on self.stream and does not wait;
        fn_side (optional, stream-ordered only) does the same on self.side_stream."""
        self._ck(self.L.nalo_ba_set_allreduce_mode(self.h_, int(bool(stream_ordered) and fn is not None)))
        if fn is None:
            self._hook = self._hook_side = None
            self._ck(self.L.nalo_ba_set_allreduce(self.h_, C.cast(None, ALLREDUCE_FN), None))
            self._ck(self.L.nalo_ba_set_allreduce_side(self.h_, C.cast(None, ALLREDUCE_FN), None))
            return
        self._hook = ALLREDUCE_FN(lambda user, ptr, n: fn(ptr, n))
        self._ck(self.L.nalo_ba_set_allreduce(self.h_, self._hook, None))
        if fn_side is not None and stream_ordered:
            self._hook_side = ALLREDUCE_FN(lambda user, ptr, n: fn_side(ptr, n))
            self._ck(self.L.nalo_ba_set_allreduce_side(self.h_, self._hook_side, None))
        else:
            self._hook_side = None
            self._ck(self.L.nalo_ba_set_allreduce_side(self.h_, C.cast(None, ALLREDUCE_FN), None))

    def ba_exchange_failed(self, what="test"):
        """what a caller's all-reduce hook calls when its collective failed (the hook has no return value)"""
        self.L.nalo_ba_exchange_failed.argtypes = [C.c_void_p, C.c_char_p]
        self._ck(self.L.nalo_ba_exchange_failed(self.h_, what.encode()))

    def ba_rccl_ranks(self):
        """(ranks of the main communicator, ranks of the side communicator) as RCCL reports them (ncclCommCount); 0 = none installed"""
        a, b = C.c_int(0), C.c_int(0)
        self._ck(self.L.nalo_ba_rccl_ranks(self.h_, C.byref(a), C.byref(b)))
        return a.value, b.value

    def ba_rccl_init(self, nranks, rank, id_main, id_side=None):
        """native RCCL exchange (ncclCommInitRank on this context's device, collective over the ranks); ids from rccl_unique_id()"""
        self._ck(self.L.nalo_ba_rccl_init(self.h_, int(nranks), int(rank), id_main, id_side))

    # ---- profiling
    def hbm_calibrate(self, nbytes=1 << 30, iters=10):
        """measured streaming bandwidth of this device (GB/s): (copy, triad) - the roofline's denominator next to the nominal 8 TB/s"""
        a, b = C.c_double(0), C.c_double(0)
        self._ck(self.L.nalo_hbm_calibrate(self.h_, C.c_size_t(nbytes), iters, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile_enable(self, on=True):
        self._ck(self.L.nalo_profile_enable(self.h_, int(on)))

    def profile_select(self, name=None):
        """bracket only the scope `name` (None = all scopes)"""
        self._ck(self.L.nalo_profile_select(self.h_, None if name is None else name.encode()))

    def profile_sample(self, every=1):
        self._ck(self.L.nalo_profile_sample(self.h_, int(every)))

    def profile_reset(self):
        self._ck(self.L.nalo_profile_reset(self.h_))

    def profile_get(self, name):
        ms, n = C.c_double(0), C.c_int(0)
        self._ck(self.L.nalo_profile_get(self.h_, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_samples(self, name):
        """every bracketed launch of the scope since the last reset, in launch order (microseconds)"""
        n = C.c_int(0)
        self._ck(self.L.nalo_profile_samples(self.h_, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.float32)
        self._ck(self.L.nalo_profile_samples(self.h_, name.encode(), _f(out), n.value, C.byref(n)))
        return out[:n.value].astype(np.float64) * 1e3
