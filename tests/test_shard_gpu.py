"""GPU test of the sharded window on ONE device: two contexts, each holding half of the active points (bench.shard), driven by two threads
whose all-reduce hook sums the buffers through the host. What the ranks agree on must be what a single context holding the whole window computes:
the stitched systems (sum of the shards), the energy, and -- exactly -- the newest frame's energy threshold (setNewFrameEnergyTH is an order statistic:
the radix histograms are summed across ranks before their searches)."""
import threading

import numpy as np
import pytest
import torch

import bench
from nalo_slam_amd import binding, synth

pytestmark = pytest.mark.gpu


class _Ptr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)


def make_ctx(win, st6):
    c = binding.Context(win.w, win.h, win.K, n_slots=win.W)
    for i in range(win.W):
        c.frame_upload(i, win.images[i])
    c.ba_set_window(list(range(win.W)), win.world_to_cam[:win.W], state6=st6)
    c.ba_set_points(win.host, win.u, win.v, win.idepth, win.color, win.weights)
    c.ba_set_residuals(win.exists)
    return c


def test_two_shards_agree_with_the_whole_window():
    win = synth.make_window(w=640, h=480, W=5, P=1500, seed=12)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    full = make_ctx(win, st6)
    e_full = full.ba_linearize()
    th_full = full.ba_get_frames()[0][win.W - 1].frameEnergyTH
    HA, bA = full.ba_accumulate(0)
    r_full = full.ba_optimize(3, never_break=True)
    th_full2 = full.ba_get_frames()[0][win.W - 1].frameEnergyTH
    w2c_full = full.ba_get_frames()[1].copy()
    full.close()

    world = 2
    bar = threading.Barrier(world)
    bufs, calls, out, err = [None] * world, [[] for _ in range(world)], [None] * world, []

    def rank_job(r):
        try:
            part = bench.shard(win, r, world)
            c = make_ctx(part, st6)

            def hook(ptr, n):                                    # blocking contract: the library drained the producing stream before the call
                t = torch.as_tensor(_Ptr(ptr, n), device="cuda")
                bufs[r] = t.cpu()
                bar.wait()
                tot = bufs[0] + bufs[1]
                bar.wait()
                t.copy_(tot)
                torch.cuda.synchronize()
                calls[r].append(n)
            c.ba_set_allreduce(hook)
            e = c.ba_linearize()
            th = c.ba_get_frames()[0][win.W - 1].frameEnergyTH
            H, b = c.ba_accumulate(0)
            c.ba_optimize(3, never_break=True)
            fr = c.ba_get_frames()
            out[r] = dict(e=e, th=th, H=H, b=b, th2=fr[0][win.W - 1].frameEnergyTH, w2c=fr[1].copy(), n=len(part.host))
            c.ba_set_allreduce(None)                             # finishes the last pass's pending level-C sum under the old hook (both ranks: it still matches)
            out[r]["e_alone"] = c.ba_linearize()                 # the context goes on as a single-GPU one (its shard only)
            c.close()
        except Exception as ex:                                  # never leave the other rank waiting in the barrier
            err.append(ex)
            bar.abort()

    ts = [threading.Thread(target=rank_job, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not err, err
    a, b = out
    assert a["n"] + b["n"] == len(win.host) and min(a["n"], b["n"]) > 0.3 * len(win.host)
    assert a["e_alone"] > 0 and b["e_alone"] > 0 and a["e_alone"] != b["e_alone"]
    # per linearisation THREE collectives: levels A and B of the threshold's radix select (1024 doubles each: 2048 bins, two per double), then
    # [systems | tail | level C] in one sum. A pass whose systems nobody fetches (the last one of optimize) sums level C alone (256) when the threshold is next needed.
    n_sys = 2 * (8 * win.W + 5) ** 2 + 2 * win.W ** 2 + 5
    sizes = calls[0]
    assert calls[0] == calls[1] and sizes[:2] == [1024, 1024] and n_sys + 256 <= sizes[2] <= n_sys + 256 + 16, sizes
    lo_off = (n_sys + 15) & ~15
    full, tail_only = lo_off + 256, lo_off - 2 * (8 * win.W + 5) ** 2 + 256     # [systems | tail | C]; optimize()'s last pass fetches [tail | C] only (energy, count, threshold)
    others = [n for n in sizes if n != 1024]
    assert sizes.count(1024) == 2 * len(others) and len(others) == 6, sizes         # six passes: nalo_ba_linearize + optimize(3) = 1 + 3 + the fixing pass; each carries C once
    assert all(n in (256, full, tail_only) for n in others) and others.count(full) >= 4 and others.count(256) <= 2, sizes
    # the threshold is the whole window's order statistic, bit for bit, on both ranks; after the optimisation too (7 passes later)
    assert a["th"] == b["th"] == th_full
    assert a["th2"] == b["th2"]
    assert abs(a["th2"] - th_full2) <= 1e-3 * th_full2           # the trajectories differ in the last digits (fp32 sums in another order), the rule is the same
    assert abs(a["e"] - e_full) < 1e-5 * e_full and a["e"] == b["e"]
    assert np.abs(a["H"] - HA).max() < 2e-5 * np.abs(HA).max() and np.array_equal(a["H"], b["H"])
    for f in range(win.W):
        d = np.abs(a["w2c"][f] - w2c_full[f]).max()
        assert d < 1e-5 and np.array_equal(a["w2c"][f], b["w2c"][f])


def test_second_fetch_of_a_pass_sums_only_what_it_stitched():
    """nalo_ba_linearize followed by nalo_ba_accumulate_sc / nalo_ba_solve_system (the step-by-step mapping of ef->solveSystemF,
    INTEGRATION.md) fetches the SAME linearisation twice. The first fetch summed H_A, b_A and the tail over the ranks; the second stitches the Schur complement
    alone and must sum ITS block alone (rounds 2-3 all-reduced the whole buffer again: world x H_A, world x b_A, world x energy / resInA). Both orders
    (top first, Schur complement first), against one context holding the whole window."""
    win = synth.make_window(w=640, h=480, W=4, P=900, seed=21)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    full = make_ctx(win, st6)
    ref = {}
    ref["e"] = full.ba_linearize()
    ref["HA"], ref["bA"] = full.ba_accumulate(0)
    ref["Hs"], ref["bs"] = full.ba_accumulate_sc(True)
    ref["x"] = full.ba_solve_system(0)
    ref["counts"] = full.ba_counts()
    full.ba_linearize()                                         # the other order, on the next linearisation (the newest frame's threshold has moved: another system)
    ref["Hs2"], _ = full.ba_accumulate_sc(True)
    ref["HA2"], _ = full.ba_accumulate(0)
    ref["x2"] = full.ba_solve_system(0)
    full.close()
    world = 2
    bar = threading.Barrier(world)
    bufs, calls, out, err = [None] * world, [[] for _ in range(world)], [None] * world, []

    def rank_job(r):
        try:
            c = make_ctx(bench.shard(win, r, world), st6)

            def hook(ptr, n):
                t = torch.as_tensor(_Ptr(ptr, n), device="cuda")
                bufs[r] = t.cpu()
                bar.wait()
                tot = bufs[0] + bufs[1]
                bar.wait()
                t.copy_(tot)
                torch.cuda.synchronize()
                calls[r].append(n)
            c.ba_set_allreduce(hook)
            o = {}
            o["e"] = c.ba_linearize()                           # A, B, [H_A | (dead SC block) | tail | C]
            o["HA"], o["bA"] = c.ba_accumulate(0)               # already stitched and summed: no collective
            o["Hs"], o["bs"] = c.ba_accumulate_sc(True)         # second fetch: the Schur complement's block alone
            o["x"] = c.ba_solve_system(0)                       # both stitched: no collective; solves with H_A and b_A as they are
            o["counts"] = c.ba_counts()
            n1 = len(calls[r])
            c.ba_linearize()                                    # the other order: Schur complement first ...
            o["Hs2"], _ = c.ba_accumulate_sc(True)
            o["HA2"], o["bA2"] = c.ba_accumulate(0)             # ... (no-op: nalo_ba_linearize stitched the top system) ...
            o["x2"] = c.ba_solve_system(0)
            o["calls2"] = calls[r][n1:]
            out[r] = o
            c.ba_set_allreduce(None)
            c.close()
        except Exception as ex:
            err.append(ex)
            bar.abort()

    ts = [threading.Thread(target=rank_job, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not err, err
    a, b = out
    blk = (8 * win.W + 5) ** 2
    n_sys = 2 * blk + 2 * win.W ** 2 + 5
    assert calls[0] == calls[1]
    first = calls[0][:len(calls[0]) - len(a["calls2"])]
    assert first[:2] == [1024, 1024] and n_sys + 256 <= first[2] <= n_sys + 272 and first[3:] == [blk], calls[0]     # the second fetch: ONE block
    assert a["calls2"][:2] == [1024, 1024] and a["calls2"][3:] == [blk], a["calls2"]
    for k in ("HA", "bA", "Hs", "bs", "x"):
        assert np.array_equal(a[k], b[k]), k                                            # both ranks hold the same sums
    sc = np.abs(ref["HA"]).max()
    assert abs(a["e"] - ref["e"]) < 1e-5 * ref["e"] and a["counts"] == ref["counts"]      # resInA: not world x
    assert np.abs(a["HA"] - ref["HA"]).max() < 2e-5 * sc and np.abs(a["bA"] - ref["bA"]).max() < 2e-5 * np.abs(ref["bA"]).max()
    assert np.abs(a["Hs"] - ref["Hs"]).max() < 2e-5 * np.abs(ref["Hs"]).max() and np.abs(a["bs"] - ref["bs"]).max() < 5e-5 * np.abs(ref["bs"]).max()
    # the step of the whole window, not of a system with world x H_A (that one is off by tens of percent); the bound is the fp32 shard-order noise of H_A - H_sc
    assert np.abs(a["x"] - ref["x"]).max() < 2e-3 * np.abs(ref["x"]).max(), np.abs(a["x"] - ref["x"]).max() / np.abs(ref["x"]).max()
    assert np.abs(a["Hs2"] - ref["Hs2"]).max() < 2e-5 * np.abs(ref["Hs2"]).max() and np.abs(a["HA2"] - ref["HA2"]).max() < 2e-5 * np.abs(ref["HA2"]).max()
    assert np.abs(a["x2"] - ref["x2"]).max() < 2e-3 * np.abs(ref["x2"]).max()


@pytest.mark.parametrize("sigma", [0.004, 0.03])
def test_energy_test_on_a_sharded_window(sigma):
    """setting_forceAceptStep = false on a SHARDED window (missing #4: rounds 2-3 returned NALO_ERR_UNSUPPORTED): every linearisation is an energy
    evaluation first and is applied only if E + E_L + E_M decreased (FullSystemOptimize.cpp:511-541). The three scalars of that test - the energy of the
    unapplied linearisation, the point part of calcLEnergyPt (EnergyFunctional.cpp:332-392), the step sums of the break test - are sums over the active points,
    so each is summed over the ranks (host_ba.hip: sum_over_ranks) and every rank takes the same branch: same accept / reject sequence and poses as ONE context
    holding the whole window; the ranks agree bit for bit."""
    win = synth.make_window(w=640, h=480, W=5, P=1500, seed=12)
    st6 = synth.perturbed_poses(win, sigma_t=sigma, sigma_r=sigma / 10)
    full = make_ctx(win, st6)
    full.set_settings(force_accept_step=False)
    r_full = full.ba_optimize(6)
    stats_full = full.ba_optimize_stats()
    w2c_full = full.ba_get_frames()[1].copy()
    full.close()
    world = 2
    bar = threading.Barrier(world)
    bufs, out, err = [None] * world, [None] * world, []

    def rank_job(r):
        try:
            c = make_ctx(bench.shard(win, r, world), st6)
            c.set_settings(force_accept_step=False)

            def hook(ptr, n):
                t = torch.as_tensor(_Ptr(ptr, n), device="cuda")
                bufs[r] = t.cpu()
                bar.wait()
                tot = bufs[0] + bufs[1]
                bar.wait()
                t.copy_(tot)
                torch.cuda.synchronize()
            c.ba_set_allreduce(hook)
            rm = c.ba_optimize(6)
            out[r] = dict(rmse=rm, stats=c.ba_optimize_stats(), w2c=c.ba_get_frames()[1].copy())
            c.ba_set_allreduce(None)
            c.close()
        except Exception as ex:
            err.append(ex)
            bar.abort()

    ts = [threading.Thread(target=rank_job, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not err, err
    a, b = out
    assert a["stats"] == b["stats"] == stats_full, (a["stats"], stats_full)                 # the same iterations, the same rejections
    assert a["rmse"] == b["rmse"] and abs(a["rmse"] - r_full) < 1e-4 * r_full
    for f in range(win.W):
        assert np.array_equal(a["w2c"][f], b["w2c"][f])
        assert np.abs(a["w2c"][f] - w2c_full[f]).max() < 3e-5
    if sigma > 0.01:
        assert stats_full[1] > 0, stats_full                                             # the large perturbation really exercises the reject branch


def test_failed_exchange_stops_the_context():
    """a collective that fails must not be ignored. The hook has no return value; it reports through nalo_ba_exchange_failed (what the built-in RCCL hooks
    do on an ncclAllReduce error, host_rccl.hip): the call that issued the hook and every later BA call of the context fail with NALO_ERR_HIP instead of solving with
    sums the other ranks never received."""
    win = synth.make_window(w=640, h=480, W=4, P=600, seed=3)
    st6 = synth.perturbed_poses(win, sigma_t=0.004, sigma_r=0.0004)
    c = make_ctx(win, st6)
    n_calls = [0]

    def hook(ptr, n):                                      # a 1-rank "sum" is the identity; the fourth call reports a failure
        n_calls[0] += 1
        if n_calls[0] == 4:
            c.ba_exchange_failed("link down (test)")
    c.ba_set_allreduce(hook)
    c.ba_linearize()                                       # three collectives, all fine
    with pytest.raises(RuntimeError, match="link down"):
        c.ba_optimize(2, never_break=True)                 # its first pass issues the fourth call
    with pytest.raises(RuntimeError, match="cross-rank sum"):
        c.ba_linearize()                                   # latched: nothing of this context runs a BA pass any more
    assert n_calls[0] == 4
    c.ba_set_allreduce(None)
    c.close()


def test_bench_gpus_2_runs_two_ranks_end_to_end(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it must run two ranks end to end: the parent starts fresh rank processes before any GPU call,
    the ranks rendezvous on 127.0.0.1, every rank runs the headline window (replicas) and its share of the sharded window with the exchange through the
    all-reduce hooks, and rank 0 prints ONE line with n_gpus = 2. One GPU here, so both ranks sit on device 0 and the exchange is the gloo rehearsal hook
    (RCCL refuses two ranks on one device): the line says so (rccl_ranks null, exchange 'python hook'); with --backend nccl on N GPUs the same code prints
    rccl_ranks = ncclCommCount and exits non-zero when it differs from N."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NALO_BENCH_ONE_DEVICE="1", NALO_BENCH_SHARD_P="40000")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launcher"] == "self" and d["scaling"] == "weak" and d["value"] > 0
    s = d["shard1m"]
    assert s["n_gpus"] == 2 and s["process_group_ranks"] == 2 and s["scaling"] == "strong" and s["keyframes_per_s"] > 0 and d["value_shard1m"] == s["keyframes_per_s"]
    assert s["rccl_ranks"] is None and "rehearsal" in s["exchange"]
    assert abs(s["points_per_rank"] - 20000) <= 8 and s["active_points"] == 40000
