#!/usr/bin/env python3
"""bench.py — keyframes/sec of the MI355X hot path (front-end tracking + sliding-window photometric BA).

One step = one keyframe on a fixed synthetic window:
  3 tracked frames x {makeImages + trackNewestCoarse}  +  setCoarseTrackingRef  +  optimize(6 GN iterations, no early exit)
  = 1 initial linearise + 6 x {accumulate A/L/SC -> stitch -> solve -> resubstitute -> step -> linearise} + 1 fixing linearise.

Workloads (config.workload):
  kitti00_8kf   BASELINE.json configs[1]: KITTI-00-shaped 1224x368, 4 pyramid levels, W=8, P=2000 active points, dense tracker
                reference (~8.7k residual inputs). Default. With N>1 ranks every GPU runs its own window (replicas, weak):
                a 2k-point window is far too small to amortise an all-reduce (north_star).
  stress250k    configs[3]: 1920x1072, 5 levels, W=8, P=250 000, R=1.75 M (Hessian-accumulation stress).
  shard1m       configs[4]: W=12, P=1 000 000 sharded over the ranks, stitched systems all-reduced over RCCL (strong).

Rank 0 prints ONE JSON line. `roofline` is measured live with HIP events on the library's stream; `cpu_baseline` times the
oracle's -O3 -march=native build (kind "port": the reference itself cannot be built here) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import nalo_pkg  # noqa: E402

nalo_pkg.load()
from nalo_slam_amd import binding, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRACKED_PER_KF = 3             # frames tracked between two keyframes (KITTI at 10 Hz: a keyframe every ~2-4 frames)

WORKLOADS = {
    "kitti00_8kf": dict(w=1224, h=368, W=8, P=2000, trk_extra=7000),
    "stress250k": dict(w=1920, h=1072, W=8, P=250000, trk_extra=0),
    "shard1m": dict(w=1920, h=1072, W=12, P=1000000, trk_extra=0),
}


class c_stdout_to_stderr:
    """librccl prints a version banner on the C-level stdout when its first communicator is made; stdout belongs to the ONE JSON line: file descriptor 1 points
    at stderr while RCCL initialises"""
    def __enter__(self):
        sys.stdout.flush()
        self.keep = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.keep, 1)
        os.close(self.keep)


def log(msg):
    print("[bench rank %s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


def make_inputs(name, seed=7):
    cfg = WORKLOADS[name]
    win = synth.make_window(w=cfg["w"], h=cfg["h"], W=cfg["W"], P=cfg["P"], seed=seed, n_extra=TRACKED_PER_KF)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    # tracker reference: residuals that target the newest keyframe (+ dense extra inputs for dense=1)
    rng = np.random.RandomState(11)
    n_in = int((win.host != win.W - 1).sum()) + cfg["trk_extra"]
    n_in = min(n_in, 300000)
    Ku = rng.uniform(5, win.w - 6, n_in).astype(np.float32)
    Kv = rng.uniform(5, win.h - 6, n_in).astype(np.float32)
    d = win.depth[win.W - 1][(Kv + 0.5).astype(int), (Ku + 0.5).astype(int)]
    ok = np.isfinite(d)
    trk = (Ku[ok], Kv[ok], (1.0 / d[ok]).astype(np.float32), (10.0 ** rng.uniform(-6, -3, ok.sum())).astype(np.float32))
    return win, st6, trk


def shard(win, rank, world):
    """Shard of the active-point set: the partition lives in the library (nalo_shard_points, include/nalo_gpu.h) — every rank gets the same
    share of EVERY host frame as a contiguous Hilbert range of the host's points, so all (h,t) bins stay populated evenly and the texels a rank gathers
    keep the reuse of the unsharded window (a block-cyclic shard made ba_linearize 1.8x less efficient per residual at N = 8)."""
    return synth.shard_window(win, rank, world)


class GpuJob:
    def __init__(self, win, st6, trk, device, hook=None):
        self.win, self.trk = win, trk
        W = win.W
        self.ctx = binding.Context(win.w, win.h, win.K, n_slots=W + TRACKED_PER_KF, device=device)
        for i in range(W + TRACKED_PER_KF):
            self.ctx.frame_upload(i, win.images[i])
        self.ctx.ba_set_window(list(range(W)), win.world_to_cam[:W], state6=st6)
        self.ctx.ba_set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
        self.ctx.ba_set_residuals(win.exists)
        if isinstance(hook, tuple):                                      # ("rccl", nranks, rank, id_main, id_side): the library's own RCCL exchange
            _, nranks, rk, id_main, id_side = hook
            with c_stdout_to_stderr():
                self.ctx.ba_rccl_init(nranks, rk, id_main, id_side)
        elif hook is not None:                                           # rehearsal path: hook = factory(ctx) -> Python all-reduce callable
            fn = hook(self.ctx)
            so = getattr(fn, "stream_ordered", False)
            # stream-ordered: a second hook on the context's side stream, so the threshold's histogram sums overlap the SC / stitch kernels
            self.ctx.ba_set_allreduce(fn, stream_ordered=so, fn_side=hook(self.ctx, True) if so else None)
        self.ctx.ba_snapshot()
        self.ctx.trk_set_ref(W - 1, *trk)
        self.T0 = [synth.se3_mul(win.world_to_cam[W + k], synth.se3_inv(win.world_to_cam[W - 1])) for k in range(TRACKED_PER_KF)]
        # constant-velocity style initial guess: 95% of the true motion (prepared once: host-side numpy is not part of the measured path)
        self.T_init = [synth_exp(orc_free_log(T) * 0.95) for T in self.T0]
        self.evals = 0
        self.count_trk_bytes, self.trk_bytes, self.trk_frames = False, 0.0, 0
        import ctypes as _C
        self._ev5, self._n5 = (_C.c_int * 5)(), (_C.c_int * 5)()
        self.last_T = [None] * TRACKED_PER_KF                           # tracked poses of the last step (pose_delta_vs_oracle)
        self._pinned = None

    def enable_raw_uploads(self):
        """page-locked 8-bit copies of the tracked frames + an identity response (photometricCalibration 1, passthrough geometry): step(upload="raw") then
        feeds the SENSOR format through nalo_frame_upload_raw_async — 1 B/px over PCIe, photometric undistortion fused in front of the pyramid"""
        W = self.win.W
        self.ctx.undist_set(self.win.w, self.win.h, np.arange(256, dtype=np.float32), None, 1, None, None)
        self._pinned_raw = []
        for k in range(TRACKED_PER_KF):
            a = self.ctx.pinned_array((self.win.h, self.win.w), np.uint8)
            a[:] = np.clip(np.rint(self.win.images[W + k]), 0, 255).astype(np.uint8)
            self._pinned_raw.append(a)
        import ctypes as C
        self._raw_ptr = [a.ctypes.data_as(C.c_void_p) for a in self._pinned_raw]        # prebuilt arguments, like the other per-keyframe calls
        self._one = C.c_float(1.0)

    def enable_uploads(self):
        """page-locked copies of the tracked frames: step(upload=True) then pays the per-frame PCIe copy through nalo_frame_upload_async (what a running
        system does per frame, FullSystem.cpp:1053-1065) instead of rebuilding the pyramid from HBM-resident irradiance"""
        W = self.win.W
        self._pinned = []
        for k in range(TRACKED_PER_KF):
            a = self.ctx.pinned_array((self.win.h, self.win.w))
            a[:] = self.win.images[W + k]
            self._pinned.append(a)

    def _prepare_calls(self):
        """ctypes arguments of the per-keyframe calls, built once: the timed loop then goes straight to the C ABI (same entry points, same
        arguments as binding.Context.{trk_track, trk_set_ref}; numpy conversions of constant inputs are not part of the measured path)"""
        import ctypes as C
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        self._T0 = [np.ascontiguousarray(T, np.float64).reshape(-1).copy() for T in self.T_init]
        self._T = np.zeros(12); self._aff = np.zeros(2); self._ref_aff = np.zeros(2); self._exp = np.ones(2, np.float32)
        self._mr = np.full(5, np.nan); self._lr = np.zeros(5); self._lf = np.zeros(3)
        self._ok, self._ne = C.c_int(0), C.c_int(0)
        self._trk_args = (self._T.ctypes.data_as(dp), self._aff.ctypes.data_as(dp), self._ref_aff.ctypes.data_as(dp), self._exp.ctypes.data_as(fp))
        self._trk_tail = (self._mr.ctypes.data_as(dp), self._lr.ctypes.data_as(dp), self._lf.ctypes.data_as(dp), C.byref(self._ok), C.byref(self._ne))
        self._ref = [np.ascontiguousarray(x, np.float32) for x in self.trk]
        self._ref_args = tuple(x.ctypes.data_as(fp) for x in self._ref)
        # the inputs of setCoarseTrackingRef are constants of the synthetic keyframe: resident in HBM before the timed region like the frames themselves (the contract's
        # "inputs already resident"); the legs that time host buffers (with_frame_uploads, with_raw_frame_uploads) hand the four arrays over per keyframe instead
        self.ctx.trk_ref_upload(*self._ref)

    def step(self, track=True, upload=False, keep=False):
        c, W, L = self.ctx, self.win.W, self.ctx.L
        if not hasattr(self, "_trk_args"):
            self._prepare_calls()
        if track and not upload:
            # a1 on the HBM-resident frames: the three makeImages passes are queued ahead of the tracking calls (they do not depend on them; with host
            # buffers the same overlap comes from nalo_frame_upload_async: the next frame's copy + pyramid run under the tracking of the current one) and ahead
            # of the harness' snapshot restore (independent of it: the device builds the pyramids while the host rewinds the window's state)
            for k in range(TRACKED_PER_KF):
                c.frame_rebuild(W + k)
        c.ba_restore()
        if track:
            if upload == "raw":
                for k in range(TRACKED_PER_KF):
                    c._ck(L.nalo_frame_upload_raw_async(c.h_, W + k, self._raw_ptr[k], 1, self._one, self._one, None))
            elif upload:                                                     # all three copies go to the copy stream at once: the first one is exposed,
                for k in range(TRACKED_PER_KF):                              # the others run under the tracking of the frame before
                    c.frame_upload_async(W + k, self._pinned[k])
            for k in range(TRACKED_PER_KF):
                self._T[:] = self._T0[k]; self._aff[:] = 0
                c._ck(L.nalo_trk_track(c.h_, W + k, *self._trk_args, c.levels - 1, *self._trk_tail))
                self.evals += self._ne.value
                if self.count_trk_bytes:                                     # untimed profile pass only: sum_l evals_l n_l 64 B of this frame's launch
                    L.nalo_trk_last_evals(c.h_, self._ev5, self._n5)
                    self.trk_bytes += 64.0 * sum(int(self._ev5[i]) * int(self._n5[i]) for i in range(5))
                    self.trk_frames += 1
                if keep:
                    self.last_T[k] = self._T.reshape(3, 4).copy()
        rm = c.ba_optimize(6, never_break=True)
        if track:
            # a2 for the new keyframe: setCoarseTrackingRef FOLLOWS the optimisation, as in makeKeyFrame (FullSystem.cpp:1404) - the next step's frames are tracked
            # against it (its inputs are constants of the synthetic keyframe here; in the reference they are the optimised window's residuals)
            if upload:
                c._ck(L.nalo_trk_set_ref(c.h_, W - 1, len(self._ref[0]), *self._ref_args))
            else:
                c._ck(L.nalo_trk_set_ref_resident(c.h_, W - 1))
        return rm


def synth_exp(xi):
    """SE(3) exp on the host (numpy), translation-first tangent."""
    R = synth.so3_exp(xi[3:])
    th = np.linalg.norm(xi[3:])
    K = np.array([[0, -xi[5], xi[4]], [xi[5], 0, -xi[3]], [-xi[4], xi[3], 0]])
    V = np.eye(3) + 0.5 * K + (K @ K) / 6.0 if th < 1e-8 else np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    return np.concatenate([R, (V @ xi[:3])[:, None]], axis=1)


def orc_free_log(T):
    """SE(3) log in numpy (no oracle on the product path)."""
    R, t = T[:, :3], T[:, 3]
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-8:
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    else:
        w = th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Vi = np.eye(3) - 0.5 * K + (K @ K) / 12.0 if th < 1e-8 else np.eye(3) - 0.5 * K + (1 - th / (2 * np.tan(th / 2))) / th ** 2 * (K @ K)
    return np.concatenate([Vi @ t, w])


def cpu_info():
    """model name, physical cores (unique physical id / core id pairs), logical CPUs and the CPUs this process may run on"""
    model, cores = None, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count()
    return dict(model=model, physical_cores=len(cores) or None, logical_cpus=os.cpu_count(), usable_cpus=aff)


def pose_delta_vs_oracle(job, win, st6, trk):
    """The second half of BASELINE.json's metric: the SAME keyframe (3 tracked frames + optimize(6)) on the strict fp32 oracle (pointwise fp32 in the
    reference's order, no FMA contraction, fp64 sums), compared with what the GPU step produced: max |log(T_gpu T_oracle^-1)| over the window's frames
    and over the tracked frames. Oracle = checker only."""
    import orc
    W = win.W
    job.step(True, keep=True)
    _, w2c_g, _ = job.ctx.ba_get_frames()
    st_g = job.ctx.ba_get_residuals()[0]
    orc.lib("f32").orc_set_sum_mode(0)
    ba = orc.ba_from_window(win, "f32", state6=st6)
    ba.set_options(nthreads=6, never_break=True)
    trkr = orc.Tracker(win.w, win.h, win.levels, win.K, "f32")
    dref = orc.make_images(win.images[W - 1], win.levels, "f32")[0]
    trkr.set_ref(dref, *trk)
    d_trk = []
    for k in range(TRACKED_PER_KF):
        dnew = orc.make_images(win.images[W + k], win.levels, "f32")[0]
        ok, T_o = trkr.track(dnew, job.T_init[k], [0, 0], [0, 0], [1, 1], win.levels - 1)[:2]
        d_trk.append(float(np.linalg.norm(orc.se3_log(synth.se3_mul(job.last_T[k], synth.se3_inv(T_o))))))
    ba.optimize(6)
    d_win = [float(np.linalg.norm(orc.se3_log(synth.se3_mul(w2c_g[f], synth.se3_inv(ba.frame(f)["worldToCam"]))))) for f in range(W)]
    st_o = ba.slots()[0]
    return dict(window_max=max(d_win), tracked_max=max(d_trk), residual_decisions_differ=int((st_g != st_o).sum()), residual_slots=int((st_o >= 0).sum()),
                oracle="strict fp32 restatement (liboracle_f32.so), same inputs, same keyframe", target="< 1e-5 (BASELINE.json)")


def cpu_baseline(win, st6, trk, budget_s=20.0, track=True, nthreads=6, linearize_mt=False):
    """The oracle's fast build (fp32 tiers, -O3 -march=native, 6 accumulate threads, single-thread linearise: the reference's
    own threading, util/NumType.h:42, FullSystemOptimize.cpp:154-164) timed on a bounded sample of the same keyframes."""
    import orc
    W = win.W
    P_cap = 20000
    if len(win.host) > P_cap:                     # bounded sample: the first P_cap points of the window, scaled back linearly
        import dataclasses
        sub = dataclasses.replace(win, host=win.host[:P_cap], u=win.u[:P_cap], v=win.v[:P_cap], idepth=win.idepth[:P_cap],
                                  idepth_true=win.idepth_true[:P_cap], color=win.color[:P_cap], weights=win.weights[:P_cap], exists=win.exists[:P_cap])
        scale = P_cap / float(len(win.host))
    else:
        sub, scale = win, 1.0
    dIs = [orc.make_images(win.images[i], 1, "fast")[0] for i in range(W)]
    t_total, n_kf = 0.0, 0
    trkr = None
    if track:
        trkr = orc.Tracker(win.w, win.h, win.levels, win.K, "fast")
        dref = orc.make_images(win.images[W - 1], win.levels, "fast")[0]
        trkr.set_ref(dref, *trk)
    while t_total < budget_s and n_kf < 50:
        ba = orc.BA(W, len(sub.host), win.w, win.h, win.K, "fast")
        for i in range(W):
            ba.set_frame(i, dIs[i], win.world_to_cam[i], state6=st6[i])
        ba.set_points(sub.host, sub.u, sub.v, sub.idepth, sub.color, sub.weights)
        ba.set_residuals(sub.exists)
        ba.prepare()
        ba.set_options(nthreads=nthreads, never_break=True)
        ba.L.orc_ba_set_linearize_mt(ba.h_, int(linearize_mt))
        t0 = time.perf_counter()
        if track:
            for k in range(TRACKED_PER_KF):
                dnew = orc.make_images(win.images[W + k], win.levels, "fast")[0]
                T0 = synth.se3_mul(win.world_to_cam[W + k], synth.se3_inv(win.world_to_cam[W - 1]))
                trkr.track(dnew, synth_exp(orc_free_log(T0) * 0.95), [0, 0], [0, 0], [1, 1], win.levels - 1)
            trkr.set_ref(dref, *trk)
        ba.optimize(6)
        t_total += time.perf_counter() - t0
        n_kf += 1
        del ba
    kfs = n_kf / t_total * scale
    sample = "%d keyframes of %s, %d of %d points%s" % (n_kf, "the same window", len(sub.host), len(win.host),
                                                       "" if scale == 1.0 else " (KF/s scaled by the point ratio)")
    note = ("reference threading: %d accumulate/resubstitute workers (NUM_THREADS, util/NumType.h:42), linearizeAll and the tracker single-threaded "
            "(FullSystemOptimize.cpp:154-164)" % nthreads) if not linearize_mt else \
           ("all usable cores: %d workers for accumulate, resubstitute AND linearizeAll (chunks of 50 points, as upstream DSO); the tracker stays "
            "single-threaded as in the reference" % nthreads)
    return dict(value=kfs, unit="keyframes/s", cores=nthreads, kind="port", sample=sample, threading=note,
                build="liboracle_fast.so: -O3 -march=native (the reference's flags, CMakeLists.txt:48), scalar calcRes, SSE (_mm_*) calcGSSSE / Accumulator9 as "
                      "MatrixAccumulators.h:1091-1166 writes them, a persistent worker pool handing out chunks of 50 points (IndexThreadReduce.h:76-137)")


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE in the environment): start N fresh rank processes, one GPU each, BEFORE this process
    makes any GPU call - it never makes one, it only waits for its children (never a re-exec of a process that has touched HIP). The ranks rendezvous on
    127.0.0.1; rank 0 prints the JSON line on the inherited stdout. A rank that dies takes the others with it (they would wait in a collective for ever)."""
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs between processes on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    log("self-launch: %d ranks on 127.0.0.1:%d (pids %s)" % (n, port, [p.pid for p in procs]))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            for i, p in enumerate(procs):                          # exactly the processes started above
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill(); rcs[i] = p.wait()
            log("self-launch: rank(s) %s failed (exit codes %s)" % (bad, rcs))
            return 1
        time.sleep(0.05)
    return 0


def launch_check(rank, world):
    """--launch-check: the rendezvous alone (gloo, CPU only): every rank joins the group, the ranks are counted with an all-reduce, a barrier-bracketed empty
    region is timed with the max over the ranks, rank 0 prints one JSON line. What tests/test_multigpu_cpu.py drives through `--gpus 2` on a box without a GPU."""
    import torch
    import torch.distributed as dist
    if os.environ.get("NALO_BENCH_TEST_DIE_RANK") == str(rank):       # test hook (tests/test_multigpu_cpu.py): a rank that dies before the rendezvous
        os._exit(7)
    if world > 1:
        dist.init_process_group(backend="gloo")
    seen = torch.ones(1)
    t = torch.zeros(1)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    t[0] = time.perf_counter() - t0
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(seen.item()), "barrier_s": float(t.item()),
                          "launcher": os.environ.get("NALO_BENCH_LAUNCHER", "external")}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if int(seen.item()) == world else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node. Without a launcher around it (no WORLD_SIZE) bench.py starts the N rank processes itself")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="kitti00_8kf", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + NALO_BENCH_ONE_DEVICE=1 rehearses the multi-rank flow on a single GPU (all ranks on device 0)")
    ap.add_argument("--no-extra", action="store_true", help="the main leg only: no pipelined / upload variants, no stress250k / shard1m / front-end legs (the command the rocprofv3 summaries are taken of)")
    ap.add_argument("--ba-only", action="store_true", help="the step is optimize() alone (no tracked frames, no setCoarseTrackingRef)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous only (gloo, no GPU): proves that --gpus N starts N ranks")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        os.environ["NALO_BENCH_LAUNCHER"] = "self"
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NALO_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus != world:
        # a line that says n_gpus: 1 for --gpus 8 (or the reverse) would be void: refuse instead of measuring something else than what was asked for
        log("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d) or drop WORLD_SIZE" % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    if args.launch_check:
        sys.exit(launch_check(rank, world))
    import torch
    if world > 1 and os.environ.get("NALO_BENCH_ONE_DEVICE") != "1" and torch.cuda.device_count() < world:
        log("%d ranks but %d visible GPUs (NALO_BENCH_ONE_DEVICE=1 with --backend gloo rehearses on one device)" % (world, torch.cuda.device_count()))
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    sharded = args.workload == "shard1m"

    win, st6, trk = make_inputs(args.workload)
    hook = rccl_setup(dist, rank, world, args.backend) if (sharded and world > 1) else None
    job = GpuJob(shard(win, rank, world) if sharded else win, st6, trk, local_rank, hook)
    main_rccl = job.ctx.ba_rccl_ranks() if isinstance(hook, tuple) else None      # --workload shard1m as the main line: its own communicators
    if main_rccl is not None and min(main_rccl) != world:
        log("rccl_ranks %s != n_gpus %d: the ranks did not join ONE communicator" % (main_rccl, world))
        sys.exit(4)
    do_track = not sharded and not args.ba_only

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        job.ctx.sync()

    for _ in range(args.warmup):
        job.step(do_track)
    # timed region: HIP events on the dominant kernel only (ba_linearize: its dispatches carry their own timestamps). Bracketing every scope with
    # recorded event pairs costs ~10 us of pipeline bubbles each on this latency-bound window (~0.2 ms per keyframe): the other kernels are
    # measured in a separate, untimed pass right after.
    mode = os.environ.get("NALO_BENCH_PROFILE", "dominant")      # dominant | all | none
    job.ctx.profile_select("ba_linearize" if mode == "dominant" else None)
    # one launch in NINE is bracketed (co-prime with the 8 linearisations of a keyframe, so every position of the loop is sampled): all eight cost 4-6 % of a
    # step on this latency-bound window, one in three 0.7 %, one in nine 0.3 % (round 4, same box: 881.9 / 885.7 / 888.3 keyframes/s for 3 / 9 / none); the stress250k leg brackets every launch
    lin_every = int(os.environ.get("NALO_BENCH_PROFILE_EVERY", "9")) if (mode == "dominant" and args.workload == "kitti00_8kf") else 1
    job.ctx.profile_sample(lin_every)
    job.ctx.profile_enable(mode != "none")
    job.ctx.profile_reset()
    job.evals = 0
    import gc
    gc.collect(); gc.disable()                                   # harness hygiene: no Python garbage collection inside the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rm = job.step(do_track)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        tt = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    job.ctx.profile_enable(False)
    units = args.steps * (1 if sharded else world)       # replicas: every rank processed its own keyframes
    value = units / dt

    prof = {"ba_linearize": job.ctx.profile_get("ba_linearize")}
    lin_stats = launch_stats(job.ctx.profile_samples("ba_linearize"), every=lin_every)
    job.ctx.profile_sample(1)
    # the roofline's denominator measured on this device in the same run: streaming copy / triad over 1 GiB buffers
    try:
        hbm_copy, hbm_triad = job.ctx.hbm_calibrate(1 << 30, 10)
    except Exception:
        hbm_copy = hbm_triad = None
    evals_timed = job.evals
    job.ctx.profile_select(None); job.ctx.profile_enable(True); job.ctx.profile_reset()
    nprof = max(2, min(5, args.steps))
    job.count_trk_bytes = True
    for _ in range(nprof):                                       # untimed: every scope bracketed
        job.step(do_track)
    job.count_trk_bytes = False
    job.ctx.profile_enable(False)
    prof.update({k: job.ctx.profile_get(k) for k in ("ba_sc", "ba_reduce", "ba_resub", "trk_eval", "trk_lm", "pyramid")})
    job.evals = evals_timed
    out = None
    cpu_legs_pending, cpu_baseline_legs = False, None          # rank 0 of a default run: the CPU baseline legs are run behind the GPU legs
    if rank == 0:
        # roofline of the dominant kernel (ba_linearize): algorithmic bytes per launch:
        #   424*R + 104*P  = R*(8 state/energy + 8px*4taps*12 B + 32 B JpJdF write) + P*(80 B point + 24 B Hdd/bd/Hcd write)
        st, ac, _, _, _ = job.ctx.ba_get_residuals()
        R = int((job.win.exists > 0).sum())
        P = len(job.win.host)
        ms, n = prof["ba_linearize"]
        roof = None
        if n > 0:
            alg = 424.0 * R + 104.0 * P
            ach = alg / (ms / n * 1e-3) / 1e9
            roof = dict(kernel="ba_linearize", bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 5), traffic=load_traffic(args.workload),
                        avg_us=round(ms / n * 1e3, 2), launches=n, alg_bytes=int(alg), stats=lin_stats,
                        measured_over="the timed region: full steps (%s), %s" % ("tracking + setCoarseTrackingRef + optimize" if do_track else "optimize only", "every launch" if lin_every == 1 else "one launch in %d" % lin_every))
            if hbm_copy:
                roof.update(measured_copy_GBs=round(hbm_copy, 1), measured_triad_GBs=round(hbm_triad, 1), frac_of_measured_copy=round(ach / hbm_copy, 4))
        out = {
            "metric": "keyframes/sec, KITTI-00 dense 8-KF photometric BA; pose RMSE vs ref",
            "value": round(value, 3), "unit": "keyframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "image": "%dx%d" % (win.w, win.h), "levels": win.levels, "window_frames": win.W,
                       "active_points": int(len(win.host)), "residuals": int((win.exists > 0).sum()),
                       "tracked_frames_per_keyframe": TRACKED_PER_KF if do_track else 0, "gn_iterations": 6,
                       "multi_gpu": ("points sharded, all-reduce of stitched H,b" if sharded else "replicas (window too small to shard)")},
            "roofline": roof,
            "kernel_ms": {k: {"total_ms": round(v[0], 3), "launches": v[1]} for k, v in prof.items()},
            "kernel_ms_note": "ba_linearize: HIP events over the timed region (%d steps%s); the other scopes: %d untimed steps right after it" % (args.steps, ", one launch in %d bracketed" % lin_every if lin_every > 1 else "", nprof),
            "tracker_evals_per_step": job.evals / max(args.steps, 1),
            "rccl_ranks": (min(main_rccl) if main_rccl else None),
            "fine_track_rmse": round(float(rm), 4),
        }
        if do_track and prof["trk_lm"][1] > 0 and job.trk_frames > 0:
            # the headline's OWN dominant kernel: the persistent LM kernel of the tracker, one launch per tracked frame. Algorithmic bytes =
            # sum over its evaluations of 64 B per point of the evaluated level; time = dispatch timestamps of the launch.
            tms, tn = prof["trk_lm"]
            alg = job.trk_bytes / job.trk_frames
            ach = alg / (tms / tn * 1e-3) / 1e9
            out["roofline_headline"] = dict(kernel="trk_lm", workload=args.workload, bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                                            traffic=load_traffic(args.workload + "_trk_lm"), avg_us=round(tms / tn * 1e3, 2), launches=tn, alg_bytes=int(alg),
                                            share_of_step=round(TRACKED_PER_KF * (tms / tn) / (dt / args.steps * 1e3), 3),
                                            note="one persistent launch per tracked frame runs the whole LM descent (%.1f evaluations): a serial chain of ~8 us per evaluation (gather 1.3 + block "
                                                 "reduction 1.8 + exchange between workgroups 2.6 + the 8x8 solve / SE3 on one wave 2.3), latency bound by construction on ~25 k points; "
                                                 "alg_bytes = sum_l evals_l n_l 64 B" % (job.evals / max(args.steps, 1) / TRACKED_PER_KF))
        if world == 1 and do_track and not args.no_extra:       # --no-extra: the main leg only (the command the rocprofv3 summaries are taken of: nothing concurrent in the trace)
            try:
                out["pipelined"] = pipelined_leg(win, st6, trk, local_rank, steps=max(20, min(200, args.steps)))
                out["value_with_uploads"] = None                             # filled below (promoted beside value)
            except Exception as e:
                out["pipelined"] = {"error": repr(e)}
            # the per-frame PCIe copy a running system pays (never part of `value`: the contract times HBM-resident inputs)
            job.enable_uploads()
            nup = max(2, min(50, args.steps))
            for _ in range(2):
                job.step(True, upload=True)
            job.ctx.sync()
            import gc
            gc.collect(); gc.disable()
            t1 = time.perf_counter()
            for _ in range(nup):
                job.step(True, upload=True)
            job.ctx.sync()
            dtu = (time.perf_counter() - t1) / nup
            gc.enable()
            out["value_with_uploads"] = round(1.0 / dtu, 3)                  # the same step fed over PCIe (float frames), promoted beside `value`
            out["with_frame_uploads"] = dict(value=round(1.0 / dtu, 3), unit="keyframes/s", ms_per_step=round(dtu * 1e3, 4), steps=nup,
                                            note="same step, but the %d tracked frames arrive through nalo_frame_upload_async from pinned host memory "
                                                 "(%.2f MB each over PCIe, copy stream overlapped with tracking) instead of HBM-resident irradiance" % (TRACKED_PER_KF, win.w * win.h * 4 / 1e6))
            try:
                job.enable_raw_uploads()
                for _ in range(2):
                    job.step(True, upload="raw")
                job.ctx.sync()
                import gc
                gc.collect(); gc.disable()                                   # the harness: a generation-2 collection of this process is a 38 ms stall (measured: one per 50 steps)
                t1 = time.perf_counter()
                for _ in range(nup):
                    job.step(True, upload="raw")
                job.ctx.sync()
                dtr = (time.perf_counter() - t1) / nup
                gc.enable()
                out["with_raw_frame_uploads"] = dict(value=round(1.0 / dtr, 3), unit="keyframes/s", ms_per_step=round(dtr * 1e3, 4), steps=nup,
                                                    note="same step, the %d tracked frames arrive as 8-bit sensor frames through nalo_frame_upload_raw_async (%.2f MB each over PCIe, "
                                                         "photometric undistortion on the device in front of makeImages; the frames are the 8-bit roundings of the synthetic images)" % (TRACKED_PER_KF, win.w * win.h / 1e6))
                for k in range(TRACKED_PER_KF):                              # back to the float frames for the legs below
                    job.ctx.frame_upload(win.W + k, win.images[win.W + k])
            except Exception as e:
                out["with_raw_frame_uploads"] = {"error": repr(e)}
        def cpu_baseline_legs():
            # ~30 s of all-core CPU work. In the default run it follows the GPU legs (round 4): the call-time figures of the legs behind it (immature points, pixel
            # selector) read 30-60 % higher inside a process that had just run it than in a fresh one (scripts/diag/imm_ctx.py); nothing of it is GPU time
            try:
                info = cpu_info()
                out["cpu_baseline"] = cpu_baseline(win, st6, trk, track=do_track)
                out["cpu_baseline"].update(cpu_model=info["model"], physical_cores=info["physical_cores"], usable_cpus=info["usable_cpus"])
                out["speedup_vs_cpu_port"] = round(value / out["cpu_baseline"]["value"], 2)
                nall = max(1, min(info["usable_cpus"] or 1, info["physical_cores"] or info["usable_cpus"] or 1, 64))
                out["cpu_baseline_all_cores"] = cpu_baseline(win, st6, trk, budget_s=10.0, track=do_track, nthreads=nall, linearize_mt=True)
                out["cpu_baseline_all_cores"].update(cpu_model=info["model"], physical_cores=info["physical_cores"], usable_cpus=info["usable_cpus"])
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            if do_track:
                out["pose_delta_vs_oracle"] = pose_delta_vs_oracle(job, win, st6, trk)
            if args.workload == "kitti00_8kf" and not args.no_extra:
                cpu_legs_pending = True                      # behind the extra GPU legs, see above
            else:
                cpu_baseline_legs()
        elif world == 1:
            out["cpu_baseline"] = None
    # extra legs of the default run (same JSON line):
    #  shard1m    configs[4]: the 1M-point 12-KF window sharded over ALL ranks with the RCCL all-reduce (strong scaling, every N)
    #  stress250k configs[3]: single GPU only, where the kernels saturate the chip
    if args.workload == "kitti00_8kf" and not args.no_extra:
        log("main leg done: %.1f keyframes/s; starting the shard1m leg" % value)
        job.ctx.close()
        # The extra legs must never cost the main line: they run under a deadline (a collective that never completes on some fabric would otherwise
        # hang every rank). On expiry rank 0 prints the line it has, marked, and every rank leaves.
        import threading
        deadline_s = float(os.environ.get("NALO_BENCH_EXTRA_DEADLINE", "420" if world == 1 else "240"))   # N > 1 runs only the sharded leg (~30 s): a blocked collective must not hold the node for long

        leg_state = {"leg": "shard1m"}                # which extra leg (hence which kernel / collective) was in flight when the deadline hit

        def _expired():
            # a leg that overruns its deadline means a hung kernel or a collective that never completed: the main line is still printed (marked),
            # but the process must not look clean to the driver: exit non-zero on every rank
            log("extra leg '%s' exceeded %.0f s: abandoning the GPU work, exit 3" % (leg_state["leg"], deadline_s))
            if rank == 0:
                snap = dict(out)                      # the main thread may be adding keys: serialise a copy
                snap["extra_legs_error"] = "leg '%s' exceeded %.0f s (hang or blocked collective); process exited with code 3" % (leg_state["leg"], deadline_s)
                print(json.dumps(snap, default=str), flush=True)
            os._exit(3)
        watchdog = threading.Timer(deadline_s, _expired)
        watchdog.daemon = True
        watchdog.start()
        try:
            res = shard_leg(rank, world, local_rank, dist, torch, backend=args.backend)
        except Exception as e:                      # never lose the main line to the extra leg
            res = {"error": repr(e)}
        log("shard1m leg: %s" % (res,))
        ranks_ok = True
        if isinstance(res, dict) and args.backend == "nccl" and "error" not in res:
            ranks_ok = res.get("rccl_ranks") == world        # every rank checks its own communicators
        if rank == 0:
            out["shard1m"] = res
            out["rccl_ranks"] = res.get("rccl_ranks") if isinstance(res, dict) else None
            out["launcher"] = os.environ.get("NALO_BENCH_LAUNCHER", "external (WORLD_SIZE set by the caller)" if "WORLD_SIZE" in os.environ else "none (single process)")
            if world > 1:                                    # N > 1: the sharded window is what the N GPUs do TOGETHER (strong scaling); promoted beside `value` (replicas, weak)
                out["value_shard1m"] = res.get("keyframes_per_s") if isinstance(res, dict) else None
                out["value_shard1m_note"] = "configs[4]: ONE 1M-point / 12-keyframe window sharded over the %d ranks, keyframes/s of the whole job (strong scaling); `value` = %d replicas of the KITTI window (weak)" % (world, world)
        if not ranks_ok:
            log("rccl_ranks %s != n_gpus %d: the ranks did not join ONE communicator; the line is void" % (res.get("rccl_ranks"), world))
            if rank == 0:
                out["rccl_ranks_error"] = "rccl_ranks != n_gpus"
                print(json.dumps(out, default=str), flush=True)
            watchdog.cancel()
            os._exit(4)
        if rank == 0 and world == 1:
            leg_state["leg"] = "stress250k"
            try:
                out["stress250k"] = stress_leg()
            except Exception as e:              # never lose the main line to an extra leg
                out["stress250k"] = {"error": repr(e)}
            log("stress250k leg done")
            leg_state["leg"] = "frontend"
            try:
                out["frontend_rooflines"] = frontend_legs()
            except Exception as e:
                out["frontend_rooflines"] = {"error": repr(e)}
            log("front-end roofline legs done")
            leg_state["leg"] = "immature"
            try:
                out["immature"] = imm_leg(cpu=not args.no_cpu_baseline)
            except Exception as e:
                out["immature"] = {"error": repr(e)}
            log("immature leg done")
            leg_state["leg"] = "pixel_selector"
            try:
                out["pixel_selector"] = pixsel_leg(cpu=not args.no_cpu_baseline)
            except Exception as e:
                out["pixel_selector"] = {"error": repr(e)}
            log("pixel selector leg done")
            leg_state["leg"] = "initializer"
            try:
                out["initializer"] = init_leg(cpu=not args.no_cpu_baseline)
            except Exception as e:
                out["initializer"] = {"error": repr(e)}
            log("initializer leg done")
            # The KITTI-sized launch moves 6 MB (0.8 us at 8 TB/s): it is launch-latency bound by construction. The kernel's
            # roofline position is therefore reported on the largest single-GPU window of this same run (configs[3]);
            # the figure of the headline workload stays next to it.
            sl = out["stress250k"].get("ba_linearize")
            if sl:
                out["roofline_kitti00_8kf"] = out["roofline"]
                out["roofline"] = dict(kernel="ba_linearize", workload="stress250k", note="the KITTI-sized launch is latency bound (see roofline_kitti00_8kf, same kernel, measured over the timed region); "
                                       "this is the same kernel on the largest single-GPU window (configs[3]) inside this run",
                                       ceiling=dict(what="pure tap gathers of the same residual list, nothing else, from the layout the kernel uses: 12-byte texels in 5x2 tiles of 128 bytes (scripts/ubench/gather.hip layout J, "
                                                         "profiles/r02_ubench_gather_12B.log; 16-byte texels row major: 175 us, in 4x2 tiles: 153 us, 12-byte row major: 152 us)",
                                                    gather_only_us=131.5, frac_of_hbm_roofline=round(768e6 / 131.5e-6 / 1e9 / HBM_PEAK_GBS, 3),
                                                    note="the gather rate saturates at 3-4 waves/SIMD (row major: 1: 252, 2: 199, 3: 180, 4: 176, 8: 175 us): bound by the miss path of sparse gathers (31 k points per 2 Mpx frame), not by latency, "
                                                         "vector ALU (IEEE vs rcp division: same time) or HBM (traffic < algorithmic bytes); the kernel's waves spend ~40 % of their life outside the gather phase, "
                                                         "which is the distance to this ceiling (DESIGN.md 3)"),
                                       bound="hbm", achieved=sl["achieved_GBs"], peak=HBM_PEAK_GBS,
                                       unit="GB/s", frac=sl["frac"], traffic=load_traffic("stress250k"), avg_us=sl["avg_us"],
                                       launches=sl["launches"], alg_bytes=sl["alg_bytes"], stats=sl.get("stats"), stats_ba_only_loop=sl.get("stats_ba_only_loop"),
                                       measured_over=sl.get("measured_over"),
                                       reproduce="python bench.py --workload stress250k --no-extra --no-cpu-baseline runs the same full steps as its main leg; its rocprofv3 --kernel-trace --stats summary is profiles/INDEX.json -> roofline")
                if hbm_copy:
                    out["roofline"].update(measured_copy_GBs=round(hbm_copy, 1), measured_triad_GBs=round(hbm_triad, 1), frac_of_measured_copy=round(sl["achieved_GBs"] / hbm_copy, 4),
                                           measured_note="nalo_hbm_calibrate in this run: copy = 2 x 1 GiB / t, triad = 3 x 1 GiB / t, 10 passes, HIP events")
        watchdog.cancel()
    if rank == 0 and cpu_legs_pending:
        cpu_baseline_legs()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
    try:
        import torch.distributed as d2
        if d2.is_initialized():
            d2.destroy_process_group()
    except Exception:
        pass


def pipelined_leg(win, st6, trk, device, steps=100, warmup=5):
    """The keyframe step as the reference THREADS it (FullSystem.cpp:175, 1183-1252: tracking on the caller's thread, mapping on its own; two CoarseTracker
    instances swapped under a mutex, FullSystem.h:310-311, FullSystem.cpp:1404-1409): three contexts on one GPU. While the mapper context optimises keyframe k and
    builds the coarse tracking reference of the tracker context `forNewKF`, the other tracker context follows the three frames towards keyframe k+1; the two
    tracker contexts swap roles every keyframe. Reported BESIDE `value` (the sequential step), never instead of it."""
    import threading
    W = win.W
    mapper = GpuJob(win, st6, trk, device)
    mapper._prepare_calls()
    trackers = []
    for _ in range(2):
        c = binding.Context(win.w, win.h, win.K, n_slots=1 + TRACKED_PER_KF, device=device)
        c.frame_upload(0, win.images[W - 1])
        for k in range(TRACKED_PER_KF):
            c.frame_upload(1 + k, win.images[W + k])
        c.trk_set_ref(0, *trk)
        trackers.append(c)
    bar = threading.Barrier(2)
    err = []
    ref = [np.ascontiguousarray(x, np.float32) for x in trk]
    import ctypes as C
    fpp = C.POINTER(C.c_float)
    ref_args = tuple(x.ctypes.data_as(fpp) for x in ref)
    dpp = C.POINTER(C.c_double)
    T0 = [np.ascontiguousarray(T, np.float64).reshape(-1).copy() for T in mapper.T_init]
    n_total = warmup + steps
    t = {}

    def tracking_thread():
        try:
            T, aff, raff, ex = np.zeros(12), np.zeros(2), np.zeros(2), np.ones(2, np.float32)
            mr, lr, lf = np.full(5, np.nan), np.zeros(5), np.zeros(3)
            ok, ne = C.c_int(0), C.c_int(0)
            for s in range(n_total):
                if s == warmup:
                    bar.wait(); t["t0"] = time.perf_counter()
                c = trackers[s % 2]
                for k in range(TRACKED_PER_KF):
                    c.frame_rebuild(1 + k)
                for k in range(TRACKED_PER_KF):
                    T[:] = T0[k]; aff[:] = 0
                    c._ck(c.L.nalo_trk_track(c.h_, 1 + k, T.ctypes.data_as(dpp), aff.ctypes.data_as(dpp), raff.ctypes.data_as(dpp), ex.ctypes.data_as(fpp), c.levels - 1,
                                             mr.ctypes.data_as(dpp), lr.ctypes.data_as(dpp), lf.ctypes.data_as(dpp), C.byref(ok), C.byref(ne)))
                bar.wait()                                               # the swap of the two trackers (FullSystem.cpp:1404-1409)
        except Exception as e:
            err.append(e); bar.abort()

    def mapping_thread():
        try:
            for s in range(n_total):
                if s == warmup:
                    bar.wait()
                c = mapper.ctx
                c.ba_restore()
                c.ba_optimize(6, never_break=True)
                nk = trackers[(s + 1) % 2]                               # coarseTracker_forNewKF: the tracker of the NEXT keyframe gets its reference here
                nk._ck(nk.L.nalo_trk_set_ref(nk.h_, 0, len(ref[0]), *ref_args))
                bar.wait()
        except Exception as e:
            err.append(e); bar.abort()

    import gc
    gc.collect(); gc.disable()
    ths = [threading.Thread(target=tracking_thread), threading.Thread(target=mapping_thread)]
    [x.start() for x in ths]
    [x.join(600) for x in ths]
    for c in trackers:
        c.sync()
    mapper.ctx.sync()
    dt = time.perf_counter() - t.get("t0", time.perf_counter())
    gc.enable()
    for c in trackers:
        c.close()
    mapper.ctx.close()
    if err:
        raise err[0]
    return dict(value=round(steps / dt, 3), unit="keyframes/s", ms_per_step=round(dt / steps * 1e3, 4), steps=steps,
                note="tracker and mapper on two host threads and three contexts of ONE GPU, as the reference threads them (FullSystem.cpp:175, 1183-1252): tracking of the %d frames "
                     "towards keyframe k+1 overlaps optimize(6) + setCoarseTrackingRef of keyframe k; the two tracker contexts swap per keyframe. NOT `value`: the contract's "
                     "step is the sequential one" % TRACKED_PER_KF)


def rccl_setup(dist, rank, world, backend="nccl"):
    """The sharded window's exchange. Default (backend nccl): the LIBRARY's own RCCL path — rank 0 draws the two ncclUniqueIds (main and side
    communicator), the process group only ships them to the other ranks; every all-reduce of the GN loop is then enqueued by libnalo_gpu.so on its own
    streams (nalo_ba_rccl_init, include/nalo_gpu.h). Backend gloo (single-GPU rehearsal): the Python hook below."""
    if backend != "nccl":
        import torch
        return lambda ctx, side=False: make_hook(dist, torch, backend, stream=ctx.side_stream if side else ctx.stream)
    with c_stdout_to_stderr():
        ids = [binding.rccl_unique_id(), binding.rccl_unique_id()] if rank == 0 else [None, None]
    if world > 1:
        dist.broadcast_object_list(ids, src=0)
    return ("rccl", world, rank, ids[0], ids[1])


def make_hook(dist, torch, backend="nccl", stream=None):
    """All-reduce hook for nalo_ba_set_allreduce: SUM n doubles in place on the device over RCCL (torch.distributed 'nccl').
    With `stream` (the library's hipStream_t) the collective is enqueued on that stream and the hook returns at once: the
    library's publish kernel follows in stream order, no host synchronisation per GN iteration (nalo_ba_set_allreduce_mode = 1)."""
    class _Ptr:                                           # wraps the library's device buffer for torch, zero copy
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)

    cache = {}
    ext = torch.cuda.ExternalStream(stream) if (stream and backend == "nccl") else None

    def hook(ptr, n):
        t = cache.get((ptr, n))
        if t is None:
            t = cache[(ptr, n)] = torch.as_tensor(_Ptr(ptr, n), device="cuda")
        if backend == "nccl":
            if ext is not None:
                with torch.cuda.stream(ext):
                    dist.all_reduce(t)                     # payload ~2*(8W+5)^2*8 B = 163 KB at W=12: latency bound over xGMI
                return
            dist.all_reduce(t)
        else:                                              # rehearsal path: stage through the host
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        torch.cuda.current_stream().synchronize()
    hook.stream_ordered = ext is not None
    return hook


def shard_leg(rank, world, local_rank, dist, torch, steps=3, warmup=1, backend="nccl"):
    """configs[4]: 1M active points, 12-KF window; every rank holds a contiguous Hilbert range of every host frame's points (nalo_shard_points), frames
    are replicated, every GN iteration all-reduces the stitched systems. Strong scaling: the window is fixed, N varies."""
    hook = rccl_setup(dist, rank, world, backend)          # N = 1 goes through a 1-rank RCCL communicator too, so the exchange path is exercised
    if os.environ.get("NALO_BENCH_SHARD_P"):               # rehearsal knob (smaller window); the judged run uses the full 1M points
        WORKLOADS["shard1m"]["P"] = int(os.environ["NALO_BENCH_SHARD_P"])
    log("shard1m: generating the %d-point window" % WORKLOADS["shard1m"]["P"])
    win, st6, trk = make_inputs("shard1m")
    emu = int(os.environ.get("NALO_BENCH_EMULATE_WORLD", "0"))        # rehearsal on one GPU: run rank 0's share of an N-rank job
    part = shard(win, 0, emu) if emu > 1 else shard(win, rank, world)
    log("shard1m: uploading %d points" % len(part.host))
    job = GpuJob(part, st6, trk, local_rank, hook)
    # the ranks the exchange REALLY spans, read back from the communicators (ncclCommCount): an N-process job whose ranks never joined one communicator must not
    # pass for an N-GPU run. Rehearsals over the Python hook have no communicator: the process group's size stands in, and the line says so.
    rccl_ranks = job.ctx.ba_rccl_ranks() if backend == "nccl" else None
    for _ in range(warmup):
        job.step(False)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    job.ctx.sync()
    t0 = time.perf_counter()                                     # timed loop: no event brackets (on a sharded window the linearisation's stop event is also what the side stream waits on)
    for _ in range(steps):
        job.step(False)
    job.ctx.sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    job.ctx.profile_select("ba_linearize")                       # the roofline kernel alone, every launch of three more keyframes (every rank runs them: the collectives stay matched)
    job.ctx.profile_enable(True)
    job.ctx.profile_reset()
    for _ in range(3):
        job.step(False)
    lin = job.ctx.profile_get("ba_linearize")
    job.ctx.profile_select(None); job.ctx.profile_reset()        # the other scopes: three more, untimed keyframes on every rank (averages over ~20 launches each;
    for _ in range(3):                                           # "ba_reduce" brackets the reduce + stitch kernels only, the threshold chain and the collectives are outside)
        job.step(False)
    job.ctx.sync()
    R, P = int((part.exists > 0).sum()), len(part.host)
    res = {"workload": "shard1m", "scaling": "strong", "n_gpus": world, "keyframes_per_s": round(steps / dt, 3),
           "ms_per_keyframe": round(dt / steps * 1e3, 3), "window_frames": win.W, "active_points": int(len(win.host)),
           "residuals": int((win.exists > 0).sum()), "points_per_rank": P, "allreduce_doubles": 2 * (8 * win.W + 5) ** 2 + 2 * win.W ** 2 + 5,
           "collectives_per_linearize": "3: levels A and B of the threshold's radix select (1024 doubles = 8 KB each, side stream, under pt_acc / SC / reduce / stitch), then [systems | tail | level C] (+256 doubles) in ONE sum on the main stream",
           "allreduce_bytes_per_linearize": 8 * (2 * 1024 + ((2 * (8 * win.W + 5) ** 2 + 2 * win.W ** 2 + 5 + 15) // 16) * 16 + 256),
           "exchange": "libnalo_gpu.so -> ncclAllReduce (RCCL) on its own streams, no host callback" if backend == "nccl" else "python hook (rehearsal)",
           "rccl_ranks": (min(rccl_ranks) if rccl_ranks else None), "rccl_ranks_main_side": (list(rccl_ranks) if rccl_ranks else None),
           "process_group_ranks": (dist.get_world_size() if dist is not None else 1)}
    ms, n = lin
    if n:
        alg = 424.0 * R + 104.0 * P
        ach = alg / (ms / n * 1e-3) / 1e9
        res["ba_linearize"] = dict(avg_us=round(ms / n * 1e3, 2), launches=n, alg_bytes=int(alg), achieved_GBs=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4))
    for k in ("ba_sc", "ba_reduce", "ba_resub"):
        ms, n = job.ctx.profile_get(k)
        if n:
            res[k + "_us"] = round(ms / n * 1e3, 2)
    job.ctx.close()
    return res


def launch_stats(samples_us, per_keyframe=8, every=1):
    """mean / spread of a kernel's bracketed launches (microseconds, launch order) and the mean by position inside a keyframe (a keyframe = 1 + 6 + 1 = 8
    linearisations; with one launch in `every` bracketed, sample i is launch i * every). What a mean alone hides: the first launch behind the tracker is cold."""
    s = np.asarray(samples_us, np.float64)
    if len(s) == 0:
        return None
    pos = (np.arange(len(s)) * every) % per_keyframe
    by_pos = [round(float(s[pos == k].mean()), 2) if (pos == k).any() else None for k in range(per_keyframe)]
    return dict(mean_us=round(float(s.mean()), 2), p50_us=round(float(np.percentile(s, 50)), 2), p95_us=round(float(np.percentile(s, 95)), 2),
                min_us=round(float(s.min()), 2), max_us=round(float(s.max()), 2), launches=int(len(s)), by_position_us=by_pos)


def load_traffic(workload):
    """HBM bytes per ba_linearize launch from the committed PMC summary (profiles/traffic_*.json), or None."""
    for name in ("traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json"):          # the newest committed measurement wins
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            try:
                v = json.load(open(p)).get(workload)
                if v is not None:
                    return v
            except Exception:
                pass
    return None


def stress_leg(steps=5, warmup=2):
    """configs[3]. keyframes_per_s: the BA alone (optimize(6): the Hessian-accumulation stress the config names). The ROOFLINE kernel is then measured over FULL
    steps of the same window - three tracked frames against the 250 k-point reference and setCoarseTrackingRef in front of every optimize, as the headline's step -
    so its mean includes the cold first launch behind the tracker (the mean of a BA-only loop is the profile's best case); every launch is
    bracketed (dispatch-attached timestamps), mean / p50 / p95 and the mean by position are reported, and `python bench.py --workload stress250k --no-extra`
    is the same loop as a command of its own (profiles/INDEX.json maps it to its rocprofv3 summary)."""
    win, st6, trk = make_inputs("stress250k")
    job = GpuJob(win, st6, trk, 0)
    for _ in range(warmup):
        job.step(False)
    job.ctx.sync()
    t0 = time.perf_counter()                                     # timed loop: no event brackets at all; the scopes are measured on the keyframes that follow
    for _ in range(steps):
        job.step(False)
    job.ctx.sync()
    dt = time.perf_counter() - t0
    job.step(True)                                               # full steps from here on
    job.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step(True)
    job.ctx.sync()
    dt_full = time.perf_counter() - t0
    job.ctx.profile_select("ba_linearize")                       # the roofline kernel alone (dispatch-attached timestamps), every launch of `steps` more FULL keyframes
    job.ctx.profile_enable(True)
    job.ctx.profile_reset()
    for _ in range(max(steps, 6)):
        job.step(True)
    lin = job.ctx.profile_get("ba_linearize")
    lin_stats = launch_stats(job.ctx.profile_samples("ba_linearize"))
    job.ctx.profile_reset()
    for _ in range(3):                                           # the same kernel in the BA-only loop, for comparison
        job.step(False)
    lin_ba_only = launch_stats(job.ctx.profile_samples("ba_linearize"))
    job.ctx.profile_select(None); job.ctx.profile_reset()        # the other scopes: a third pass
    for _ in range(2):
        job.step(False)
    job.ctx.sync()
    R, P = int((win.exists > 0).sum()), len(win.host)
    res = {"workload": "stress250k", "keyframes_per_s": round(steps / dt, 3), "ms_per_keyframe": round(dt / steps * 1e3, 3),
           "active_points": P, "residuals": R, "note": "keyframes_per_s: BA only (optimize), no front-end; keyframes_per_s_full_step: 3 tracked frames (250 k-point reference) + setCoarseTrackingRef + optimize",
           "keyframes_per_s_full_step": round(steps / dt_full, 3), "ms_per_keyframe_full_step": round(dt_full / steps * 1e3, 3)}
    for k, alg in (("ba_linearize", 424.0 * R + 104.0 * P), ("ba_sc", 32.0 * R + 56.0 * P), ("ba_resub", 32.0 * R + 24.0 * P)):
        ms, n = lin if k == "ba_linearize" else job.ctx.profile_get(k)
        if n:
            ach = alg / (ms / n * 1e-3) / 1e9
            res[k] = dict(avg_us=round(ms / n * 1e3, 2), launches=n, alg_bytes=int(alg), achieved_GBs=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4))
            if k == "ba_sc":
                # what the fused kernel really moves (counter traffic 125 MB at these sizes): 32 B of Jacobian products + the 24-B point-sum share ba_linearize leaves per
                # residual + one more byte of state, and 116 B per point (the 56 above + the per-point sums written and the operands staged): the SURVEY 8(d) figure
                # above prices the shares at zero
                moved = 57.0 * R + 116.0 * P
                res[k].update(moved_bytes=int(moved), frac_moved=round(moved / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    if "ba_linearize" in res:
        res["ba_linearize"].update(measured_over="full steps (tracking + setCoarseTrackingRef + optimize), every launch", stats=lin_stats, stats_ba_only_loop=lin_ba_only)
    res["traffic"] = load_traffic("stress250k")
    job.ctx.close()
    return res


def frontend_legs(rounds=30):
    """Roofline legs of the front-end / map kernels on the 1920x1072, 5-level stress frame, each with its ALGORITHMIC bytes:
      trk_eval   fused calcRes + calcGSSSE (CoarseTracker.cpp:891-1049, 828-885) at level 0: 64 B per point (16 B point + four 12-B taps), at the
                 n0 = 250 000 points of the spec and at full density (every level-0 pixel a point: the kernel's asymptote)
      pyramid    makeImages (HessianBlocks.cpp:127-190): 4wh in + 16 wh sum_l 4^-l out  (~25.3 B/px)
      dense_map  DenseMapping::makeMap (MapPoint.cpp:334-407): 11 B per bbox pixel + 24 B per output point; dense_bbox: 4 B per scanned pixel
    Times are HIP events on the library's stream around the kernels of each scope."""
    cfg = WORKLOADS["stress250k"]
    w, h = cfg["w"], cfg["h"]
    win = synth.make_window(w=w, h=h, W=2, P=64, seed=7, n_extra=0)
    scene = synth.Scene(20240601)
    xi = np.array([0.2, 0.0, 0.5, 0.0, 0.004, 0.0])                       # new frame = reference pose o exp(xi)
    T_rel = synth_exp(xi)
    img_new, _ = scene.render(w, h, win.K, synth.se3_mul(T_rel, win.world_to_cam[0]))
    mask = np.zeros((h, w), np.float32)
    mask[600:1060, 100:1800] = 7.0                                        # a ground cluster of the mp-mask
    bgr = np.repeat(np.clip(win.images[0], 0, 255).astype(np.uint8)[:, :, None], 3, axis=2)
    c = binding.Context(w, h, win.K, n_slots=2)
    c.frame_upload(0, win.images[0], mask=mask, bgr=bgr)
    c.frame_upload(1, img_new)
    res = {"image": "%dx%d" % (w, h), "levels": c.levels}

    def timed(scope, fn, alg_bytes, n_rounds=rounds):
        fn()
        c.profile_select(scope); c.profile_enable(True); c.profile_reset()
        for _ in range(n_rounds):
            fn()
        ms, n = c.profile_get(scope)
        c.profile_enable(False)
        us = ms / max(n, 1) * 1e3
        ach = alg_bytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
        return dict(avg_us=round(us, 2), launches=n, alg_bytes=int(alg_bytes), achieved_GBs=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4), bound="hbm", peak=HBM_PEAK_GBS)

    # ---- tracker evaluation at level 0
    rng = np.random.RandomState(7)
    d0 = win.depth[0]
    for name, n_pts in (("trk_eval_250k", 250000), ("trk_eval_full_density", None)):
        if n_pts is None:
            vv, uu = np.mgrid[4:h - 4, 4:w - 4]
            u, v = uu.reshape(-1), vv.reshape(-1)
        else:
            idx = np.unique(rng.randint(4, h - 4, 2 * n_pts) * w + rng.randint(4, w - 4, 2 * n_pts))
            idx = np.sort(rng.permutation(idx)[:n_pts])                  # raster order, like the compaction of makeCoarseDepthL0 (CoarseTracker.cpp:493-538)
            u, v = idx % w, idx // w
        ok = np.isfinite(d0[v, u])
        u, v = u[ok], v[ok]
        c.trk_set_pc(0, 0, u.astype(np.float32), v.astype(np.float32), (1.0 / d0[v, u]).astype(np.float32), win.images[0][v, u].astype(np.float32))
        st = [None]

        def ev():
            st[0] = c.trk_eval(1, 0, T_rel, [1.0, 0.0], 0.0, 20.0)[0]
        r = timed("trk_eval", ev, 64.0 * len(u))
        r.update(points=int(len(u)), residuals_in=int(st[0][1]), note="64 B/point = 16 B point record + four 12-B bilinear taps counted uncached; avg_us holds the evaluation AND its fp64 finish "
                 "(one launch since round 4: the last workgroup sums the block partials; round 3 reported the evaluation alone, 14.0 us, beside a 9.9 us finish launch): %s" % ("spec size, 16 MB: 2 us at 8 TB/s, launch-latency bound by construction" if n_pts else "asymptote of the kernel"))
        res[name] = r
    # ---- pyramid
    L = c.levels
    alg_pyr = 4.0 * w * h + 16.0 * w * h * sum(0.25 ** l for l in range(L))
    res["pyramid"] = timed("pyramid", lambda: c.frame_rebuild(1), alg_pyr)
    res["pyramid"]["note"] = "ONE launch: fine tiles (64x16, LDS with halo) for the levels 0-2, coarse workgroups of the same launch rebuild the levels 3-4 from level 0; 4wh in + 16 B per pixel of every level out"
    # ---- dense map
    import ctypes as C
    cap = w * h
    plane = np.array([0.0, 1.0, 0.0, -1.6], np.float32)
    c2w = np.ascontiguousarray(np.concatenate([np.eye(3), np.zeros((3, 1))], 1)).reshape(-1)
    rect = np.zeros(4, np.int32)
    ou, ov = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    oid, oc, ob = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((cap, 3), np.uint8)
    n, acc = C.c_int(0), C.c_int(0)

    def dm():
        c._ck(c.L.nalo_dense_make_map(c.h_, 0, binding._f(plane), 7.0, binding._d(c2w), cap, binding._i(rect), binding._i(ou), binding._i(ov),
                                      binding._f(oid), binding._f(oc), binding._u8(ob), C.byref(n), C.byref(acc)))
    dm()
    bbox_px = int((rect[1] - rect[0]) * (rect[3] - rect[2]))
    res["dense_map"] = timed("dense_map", dm, 11.0 * bbox_px + 24.0 * n.value, n_rounds=10)
    res["dense_map"].update(bbox_px=bbox_px, points_out=int(n.value), note="row extents (dense_bbox scope) + two launches (chunk counts, ordered write; kernels_dense.hip) over the candidate pixels (i % 3 == 0 || j % 3 == 0) of the box, "
                            "no atomics per lane and no look-back: every workgroup folds the chunk aggregates in front of it; the accept test of MapPoint.cpp:403 (the reference's order-dependent maxy / maxz) rides along; the D2H copy of the point list is not in avg_us")
    res["dense_bbox"] = timed("dense_bbox", dm, 4.0 * (w - 4) * (h - 4), n_rounds=10)
    # ---- raw-frame ingest (photometric undistortion + remap fused in front of makeImages): an 8-bit sensor frame slightly larger than the rectified image
    wo, ho = w + 64, h + 48
    raw = rng.randint(0, 256, (ho, wo)).astype(np.uint8)
    G = np.linspace(0, 255, 256).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    rx, ry = (xx * np.float32((wo - 2.0) / w) + 0.3).astype(np.float32), (yy * np.float32((ho - 2.0) / h) + 0.3).astype(np.float32)
    vinv = np.ones((ho, wo), np.float32)
    c.undist_set(wo, ho, G, vinv, 2, rx, ry)
    alg_ing = 1.0 * wo * ho + 4.0 * wo * ho + 8.0 * w * h + 4.0 * w * h          # raw bytes + vignette + remap tables + irradiance out
    res["ingest"] = timed("ingest", lambda: c.frame_upload_raw(1, raw, exposure=0.02), alg_ing, n_rounds=10)
    res["ingest"]["note"] = ("nalo_frame_upload_raw: G[raw] * vignetteMapInv at the four taps of every rectified pixel (bilinear remap), one pass; the frame crosses PCIe "
                             "at 1 B/px (%.2f MB) instead of 4 B/px; avg_us is the kernel, the call also pays the copy and the pyramid" % (wo * ho / 1e6))
    c.close()
    # HBM traffic per launch from the committed counter passes (profiles/traffic_r04.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of scripts/diag/frontend_prof.py)
    for leg, keys in (("pyramid", ("fe_pyramid",)), ("ingest", ("fe_ingest",)), ("dense_map", ("fe_dense_count", "fe_dense_write")),
                      ("trk_eval_250k", ("fe_trk_eval",)), ("trk_eval_full_density", ("fe_trk_eval",))):
        if isinstance(res.get(leg), dict):
            t = [load_traffic(k2) for k2 in keys]
            res[leg]["traffic"] = None if any(v is None for v in t) else int(sum(t))
            if leg.startswith("trk_eval"):
                res[leg]["traffic_note"] = "counter mean over BOTH tracker-evaluation legs of the profiled script (250 k points and full density)"
    return res


def init_leg(w=1224, h=368, frames=3, cpu=True):
    """the two-frame initialiser behind the C-ABI (setFirst + trackFrame) on the KITTI frame shape, wall clock per call, next to the CPU port"""
    import time
    from nalo_slam_amd import binding, synth
    win = synth.make_window(w=w, h=h, W=2, P=20, seed=3, n_extra=frames - 1, step_z=0.15, yaw_deg=0.1)
    rp = None
    try:
        import orc
        rp, _ = orc.pixsel_libc_tables(w * h)
    except Exception:
        rng = np.random.RandomState(1); rp = rng.randint(0, 256, w * h).astype(np.uint8)
    c = binding.Context(w, h, win.K, n_slots=frames + 1)
    for i in range(frames + 1):
        c.frame_upload(i, win.images[i])
    c.pixsel_set_random(rp)
    c.init_set_first(0)
    t0 = time.perf_counter(); num, _ = c.init_set_first(0); t_first = time.perf_counter() - t0
    ev0 = c.init_state()["n_evals"]; t_tr = []
    for i in range(1, frames + 1):
        t0 = time.perf_counter(); c.init_track_frame(i); t_tr.append(time.perf_counter() - t0)
    evals = c.init_state()["n_evals"] - ev0
    res = {"image": "%dx%d" % (w, h), "points_per_level": [int(x) for x in num], "set_first_ms": round(t_first * 1e3, 2), "track_frame_ms": round(float(np.mean(t_tr)) * 1e3, 2),
           "calc_res_and_gs_evaluations_per_frame": round(evals / frames, 1),
           "note": "setFirst = selection kernels of every level + the k-d trees on the host (one level per thread, queries on 8 threads); trackFrame = ~35 evaluations on device-resident "
                   "point arrays, 94 doubles down per evaluation; optReg / resetPoints run as dependency-ordered device sweeps, bit-equal to the sequential ones"}
    c.close()
    if cpu:
        import orc
        ini = orc.Initializer(w, h, win.levels, win.K, "fast")
        t0 = time.perf_counter(); ini.set_first(win.images[0], rp); res["set_first_cpu_port_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        t = []
        for i in range(1, frames + 1):
            t0 = time.perf_counter(); ini.track_frame(win.images[i]); t.append(time.perf_counter() - t0)
        res["track_frame_cpu_port_ms"] = round(float(np.mean(t)) * 1e3, 2)
        res["cpu_cores"] = 1
    return res


def imm_leg(per_host=1500, rounds=20, cpu=True):
    """traceOn for setting_desiredImmatureDensity = 1500 immature points per host x 8 hosts against a new KITTI-sized
    frame, then optimizeImmaturePoint for the same points. Kernel time from HIP events, call time includes the PCIe staging of the
    caller-owned point arrays; the CPU figure is the oracle's fast build on one thread (the reference chunks 50 points per thread)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from imm_helpers import imm_points, host_to_new, true_idepth
    cfg = WORKLOADS["kitti00_8kf"]
    win = synth.make_window(w=cfg["w"], h=cfg["h"], W=cfg["W"], P=64, seed=9, n_extra=1, step_z=0.25, yaw_deg=0.4)
    W = win.W
    c = binding.Context(win.w, win.h, win.K, n_slots=W + 1)
    for i in range(W + 1):
        c.frame_upload(i, win.images[i])
    u, v, host = imm_points(win, per_host=per_host, seed=2)
    n = len(u)
    color, weights, gradH, eth = [np.zeros((n, k), np.float32) for k in (8, 8, 3)] + [np.zeros(n, np.float32)]
    for h in range(W):
        m = host == h
        color[m], weights[m], gradH[m], eth[m] = c.imm_create(h, u[m], v[m])
    uf, vf = u.astype(np.float32), v.astype(np.float32)
    KRKi, Kt, aff = host_to_new(win, W)
    st0 = (np.zeros(n, np.float32), np.full(n, np.nan, np.float32), np.full(n, 5, np.int32), np.full(n, 10000, np.float32))
    c.imm_trace(W, uf, vf, color, weights, gradH, eth, host, KRKi, Kt, aff, *st0)               # warm-up
    c.profile_enable(True); c.profile_reset()
    t0 = time.perf_counter()
    for _ in range(rounds):
        g = c.imm_trace(W, uf, vf, color, weights, gradH, eth, host, KRKi, Kt, aff, *st0)
    t_call = (time.perf_counter() - t0) / rounds
    ms, nl = c.profile_get("imm_trace")
    # device-resident form: the set is uploaded once, every frame only sends its 14 floats per host and nothing is waited for
    c.imm_resident_set(uf, vf, color, weights, gradH, eth, host, *st0)
    c.imm_resident_trace(W, KRKi, Kt, aff); c.sync()
    t0 = time.perf_counter()
    for _ in range(rounds):
        c.imm_resident_trace(W, KRKi, Kt, aff)
    c.sync()
    t_res = (time.perf_counter() - t0) / rounds
    t0 = time.perf_counter()
    for _ in range(rounds):                                     # the same call, completion of every frame's trace waited for (a caller that needs the result per frame)
        c.imm_resident_trace(W, KRKi, Kt, aff); c.sync()
    t_res_sync = (time.perf_counter() - t0) / rounds
    t0 = time.perf_counter()
    c.imm_resident_get()
    t_get = time.perf_counter() - t0
    idt = true_idepth(win, u, v, host)
    c.ba_set_window(list(range(W)), win.world_to_cam[:W])
    c.imm_optimize(host, uf, vf, color, weights, eth, idt * 0.9, idt * 1.1, 1)
    c.profile_reset()
    t0 = time.perf_counter()
    for _ in range(rounds):
        ro = c.imm_optimize(host, uf, vf, color, weights, eth, idt * 0.9, idt * 1.1, 1)
    t_opt = (time.perf_counter() - t0) / rounds
    ms2, nl2 = c.profile_get("imm_optimize")
    # the same activation on the device-resident set (round 4): the points are named by index, their intervals are the ones the device holds
    st1 = ((idt * 0.9).astype(np.float32), (idt * 1.1).astype(np.float32), np.full(n, 5, np.int32), np.full(n, 10000, np.float32))
    c.imm_resident_set(uf, vf, color, weights, gradH, eth, host, *st1)
    rr = c.imm_resident_optimize(None, 1, n_all=n)
    t0 = time.perf_counter()
    for _ in range(rounds):
        rr = c.imm_resident_optimize(None, 1, n_all=n)
    t_opt_res = (time.perf_counter() - t0) / rounds
    same_opt = bool(np.array_equal(rr[0], ro[0]) and np.array_equal(rr[1], ro[1], equal_nan=True) and np.array_equal(rr[2], ro[2]))
    c.close()
    res = {"points": n, "hosts": W, "image": "%dx%d" % (win.w, win.h),
           # trace_call_us = the DEFAULT path: the device-resident set (uploaded once per keyframe by makeNewTraces / activation), one call per frame,
           # completion waited for; trace_resident_us_per_frame = the same calls queued without waiting; trace_staged_call_us = the one-call form that moves all
           # 30 floats per point of the caller-owned arrays both ways on every call (rounds 1-3 reported that one as trace_call_us)
           "trace_kernel_us": round(ms / max(nl, 1) * 1e3, 1), "trace_call_us": round(t_res_sync * 1e6, 1), "trace_staged_call_us": round(t_call * 1e6, 1),
           "trace_Mpoints_per_s_kernel": round(n / (ms / max(nl, 1) * 1e-3) / 1e6, 2),
           "trace_resident_us_per_frame": round(t_res * 1e6, 1), "resident_get_us": round(t_get * 1e6, 1),
           "trace_status_counts": np.bincount(g[2], minlength=6).tolist(),
           # optimize_call_us = the device-resident set (nalo_imm_resident_optimize: 4 B per point down, results up); optimize_staged_call_us = the one-call form that
           # moves the caller-owned arrays (88 B per point) on every call (rounds 1-3 reported that one as optimize_call_us)
           "optimize_kernel_us": round(ms2 / max(nl2, 1) * 1e3, 1), "optimize_call_us": round(t_opt_res * 1e6, 1), "optimize_staged_call_us": round(t_opt * 1e6, 1),
           "optimize_resident_equals_staged": same_opt, "activated": int((ro[0] == 1).sum())}
    if cpu:
        import orc
        dI = [orc.make_images(win.images[i], 1, "fast")[0] for i in range(W + 1)]
        t0 = time.perf_counter()
        orc.imm_trace(dI[W], win.w, win.h, uf, vf, color, weights, gradH, eth, host, KRKi, Kt, aff, *st0, kind="fast")
        res["trace_cpu_port_us"] = round((time.perf_counter() - t0) * 1e6, 1)
        res["trace_cpu_cores"] = 1
    return res


def pixsel_leg(rounds=50, cpu=True):
    """PixelSelector::makeMaps (makeHists + select + sub-selection, the per-keyframe candidate selection of makeNewTraces) on a
    KITTI-sized frame, setting_desiredImmatureDensity = 1500. Call time = everything between the C-ABI call and the filled w*h float map on the host
    (what the reference's caller gets); kernel time from HIP events. The CPU figure is the oracle's fast build on one thread."""
    cfg = WORKLOADS["kitti00_8kf"]
    win = synth.make_window(w=cfg["w"], h=cfg["h"], W=2, P=64, seed=9, n_extra=0)
    import orc
    w, h = win.w, win.h
    rp, draws = orc.pixsel_libc_tables(w * h)
    c = binding.Context(w, h, win.K, n_slots=1)
    c.frame_upload(0, win.images[1])
    c.pixsel_set_random(rp, draws)
    m, num, pot = c.pixsel_make_maps(0, 1500.0, 3)               # warm-up; also adapts the potential like the running system
    c.profile_enable(True); c.profile_reset()
    t0 = time.perf_counter()
    for _ in range(rounds):
        c.frame_rebuild(0)                                       # a new frame: the block thresholds are recomputed, as per keyframe
        m, num, pot2 = c.pixsel_make_maps(0, 1500.0, pot)
    t_call = (time.perf_counter() - t0) / rounds
    ms, nl = c.profile_get("pixsel")
    c.profile_reset()
    t0 = time.perf_counter()
    for _ in range(rounds):
        c.frame_rebuild(0)
    t_rebuild = (time.perf_counter() - t0) / rounds
    c.close()
    res = {"image": "%dx%d" % (w, h), "density": 1500, "potential": pot, "selected": num,
           "make_maps_call_us": round((t_call - t_rebuild) * 1e6, 1), "make_maps_gpu_us": round(ms / max(nl, 1) * 1e3, 1)}
    if cpu:
        dI, ab = orc.make_images(win.images[1], 3, "fast")
        o1, o2 = w * h, w * h + (w // 2) * (h // 2)
        t0 = time.perf_counter()
        for _ in range(3):
            mo, numo, _ = orc.pixsel_make_maps(dI[:o1], ab[:o1], ab[o1:o2], ab[o2:], w, h, rp, 1500.0, pot, 1, 1.0, kind="fast")
        res["make_maps_cpu_port_us"] = round((time.perf_counter() - t0) / 3 * 1e6, 1)
        res["cpu_cores"] = 1
        res["same_selection_as_cpu_port"] = bool(numo == num and np.array_equal(mo, m))
    return res


if __name__ == "__main__":
    main()
