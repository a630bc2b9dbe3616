"""GPU parity: pyramid (a1), coarse depth (a2), fused calcRes+calcGS (a3+a4), LM tracking — HIP path through the
C-ABI vs the CPU oracle on identical seeded inputs. Tolerances are stated per check."""
import numpy as np
import pytest

import orc
from helpers import rel_err, tracker_inputs, true_rel_pose, pose_dist
from nalo_slam_amd import binding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(small_window):
    win = small_window
    c = binding.Context(win.w, win.h, win.K, n_slots=win.W + 1)
    for i in range(win.W + 1):
        c.frame_upload(i, win.images[i])
    yield c
    c.close()


def test_pyramid_bit_exact(ctx, small_window):
    """a1: box-mean pyramid + flat-index central differences: same fp32 operations in the same order -> bit exact."""
    win = small_window
    dI_ref, ab_ref = orc.make_images(win.images[2], win.levels)
    L = orc.lib()
    for l in range(win.levels):
        dI, ab = ctx.frame_download(2, l)
        o, n = L.orc_pyr_offset(win.w, win.h, l), (win.w >> l) * (win.h >> l)
        assert np.array_equal(dI, dI_ref[o:o + n]), "level %d texels differ" % l
        assert np.array_equal(ab, ab_ref[o:o + n])


def test_coarse_depth_and_point_clouds(ctx, small_window):
    """a2: scatter + sum pyramid + dilation + ordered compaction. Same raster order, values within 1 ulp-ish (1e-6 rel):
    the scatter sums at most a handful of terms per pixel."""
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win)
    trk = orc.Tracker(win.w, win.h, win.levels, win.K)
    dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    ctx.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi)
    for l in range(win.levels):
        a = trk.get_pc(l)
        b = ctx.trk_get_pc(l)
        assert len(a[0]) == len(b[0]) and len(a[0]) > 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])          # u, v: identical order
        assert rel_err(b[2], a[2]) < 1e-6 and np.array_equal(a[3], b[3])
        ia, wa = trk.get_depth(l)
        ib, wb = ctx.trk_get_depth(l)
        assert rel_err(ib, ia) < 1e-6 and rel_err(wb, wa) < 1e-6


def test_resident_reference_inputs_equal_the_host_arrays(small_window):
    """nalo_trk_ref_upload + nalo_trk_set_ref_resident (round 4: the per-keyframe call without host copy, staging and copy packet): clouds and depth maps of every level
    are those of nalo_trk_set_ref on the same arrays bit for bit, repeatedly (other users of the shared staging in between), and the call refuses to run without an upload"""
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win)
    a = binding.Context(win.w, win.h, win.K, n_slots=win.W + 1)
    b = binding.Context(win.w, win.h, win.K, n_slots=win.W + 1)
    for c in (a, b):
        for i in range(win.W + 1):
            c.frame_upload(i, win.images[i])
    with pytest.raises(RuntimeError):
        b.trk_set_ref_resident(win.W - 1)
    a.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi)
    b.trk_ref_upload(Ku, Kv, nid, hdi)
    for rep in range(3):
        b.trk_set_ref_resident(win.W - 1)
        for l in range(win.levels):
            pa, pb = a.trk_get_pc(l), b.trk_get_pc(l)
            assert all(np.array_equal(x, y) for x, y in zip(pa, pb)), "level %d (repeat %d)" % (l, rep)
            da, db = a.trk_get_depth(l), b.trk_get_depth(l)
            assert np.array_equal(da[0], db[0]) and np.array_equal(da[1], db[1])
        b.frame_upload(win.W, win.images[win.W])                       # the shared staging buffers are used by someone else in between
        b.trk_set_ref(win.W - 1, Ku[:100], Kv[:100], nid[:100], hdi[:100])   # and the host-array call leaves the resident block alone
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.9)
    a.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi); b.trk_set_ref_resident(win.W - 1)
    ra, rb = a.trk_track(win.W, T0, [0, 0], [0, 0], [1, 1], win.levels - 1), b.trk_track(win.W, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)
    assert ra[0] == rb[0] and np.array_equal(np.asarray(ra[1]), np.asarray(rb[1]))
    a.close(); b.close()


def test_scatter_collisions_are_summed_in_residual_order(ctx, small_window):
    """a2 step 1 with MANY residuals on the same pixels (3 .. 40 hits): the reference adds them serially in residual order; the device redoes every pixel with
    >= 3 hits in ascending index (trk_scatter_fix_kernel), so level 0's idepth / weightSums equal the oracle's BIT FOR BIT and repeat exactly"""
    win = small_window
    rng = np.random.RandomState(9)
    Ku, Kv, nid, hdi = tracker_inputs(win, n=2000, seed=3)
    hot_u, hot_v = rng.randint(20, win.w - 20, 60), rng.randint(20, win.h - 20, 60)
    reps = rng.randint(3, 41, 60)
    eu = np.concatenate([np.full(r, u) + rng.uniform(-0.45, 0.45, r) for u, r in zip(hot_u, reps)]).astype(np.float32)
    ev = np.concatenate([np.full(r, v) + rng.uniform(-0.45, 0.45, r) for v, r in zip(hot_v, reps)]).astype(np.float32)
    en = (10.0 ** rng.uniform(-2.5, 0.5, len(eu))).astype(np.float32)            # inverse depths over three decades: the order of the fp32 adds matters
    eh = (10.0 ** rng.uniform(-7, -2, len(eu))).astype(np.float32)
    perm = rng.permutation(len(Ku) + len(eu))
    Ku, Kv, nid, hdi = [np.concatenate([a, b])[perm] for a, b in ((Ku, eu), (Kv, ev), (nid, en), (hdi, eh))]
    trk = orc.Tracker(win.w, win.h, win.levels, win.K)
    dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    ia, wa = trk.get_depth(0)
    runs = []
    for _ in range(3):
        ctx.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi)
        runs.append(ctx.trk_get_depth(0))
    for ib, wb in runs:
        assert np.array_equal(ib, ia) and np.array_equal(wb, wa)
    for l in range(win.levels):
        a, b = trk.get_pc(l), ctx.trk_get_pc(l)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and rel_err(b[2], a[2]) < 1e-6


@pytest.mark.parametrize("lvl", [0, 1, 3])
def test_fused_eval_matches_oracle(ctx, small_window, lvl):
    """a3+a4: stats6, H (8x8), b (8). fp32 pointwise, fp32 block partials, fp64 finish vs the fp64-sum oracle:
    tolerance 2e-5 relative to max|H| (SURVEY §8d expects <= 1e-5 of ||H||max for fp32 partials)."""
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win)
    trk = orc.Tracker(win.w, win.h, win.levels, win.K)
    dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels)
    dI_new, _ = orc.make_images(win.images[win.W], win.levels)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    ctx.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi)
    T = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.9)
    aff = np.array([0.98, 1.5], np.float32)
    orc.lib().orc_set_sum_mode(0)
    st_o = trk.calc_res(dI_new, lvl, T, aff, 20.0)
    H_o, b_o = trk.calc_gs(lvl, float(aff[0]), 0.3)
    st, H, b = ctx.trk_eval(win.W, lvl, T, aff, 0.3, 20.0)
    assert st[1] == st_o[1] and st[1] > 100                                        # same inlier count
    assert abs(st[0] - st_o[0]) / st_o[0] < 2e-6
    assert abs(st[5] - st_o[5]) < 1e-7
    if lvl == 0:
        assert rel_err(st[[2, 4]], st_o[[2, 4]]) < 1e-5
    assert rel_err(H, H_o) < 2e-5 and rel_err(b, b_o) < 2e-5
    assert np.abs(H - H.T).max() <= 1e-12 * np.abs(H).max()
    # the reference-order fp32 oracle (SSE lanes + 3-tier) sits inside the same band
    orc.lib().orc_set_sum_mode(1)
    trk.calc_res(dI_new, lvl, T, aff, 20.0)
    H_r, b_r = trk.calc_gs(lvl, float(aff[0]), 0.3)
    orc.lib().orc_set_sum_mode(0)
    assert rel_err(H_r, H_o) < 2e-5


def test_track_recovers_pose_and_matches_oracle(ctx, small_window):
    """trackNewestCoarse: LM on the host driving the fused kernel. Pose parity vs oracle: |log(T_gpu^-1 T_ref)| < 1e-5
    (BASELINE.json target); also recovers the synthetic ground truth to < 2e-3."""
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win)
    trk = orc.Tracker(win.w, win.h, win.levels, win.K)
    dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels)
    dI_new, _ = orc.make_images(win.images[win.W], win.levels)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    ctx.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi)
    Ttrue = true_rel_pose(win, win.W - 1, win.W)
    T0 = orc.se3_exp(orc.se3_log(Ttrue) * 0.8)
    ok_o, T_o, aff_o, lr_o, _ = trk.track(dI_new, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)
    ok, T, aff, lr, lf, nev = ctx.trk_track(win.W, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)
    assert ok == 1 and ok_o == 1 and nev > 4
    assert pose_dist(T, T_o) < 1e-5
    assert np.abs(aff - aff_o).max() < 1e-3
    assert pose_dist(T, Ttrue) < 2e-3
    assert np.allclose(lr[:win.levels], lr_o[:win.levels], rtol=1e-4)


def test_track_abort_and_exposure_paths(small_window):
    """trackNewestCoarse's early exit on minResForAbort (CoarseTracker.cpp:1227-1229: outputs untouched, returns false) and the affine-brightness path with
    unequal exposures (AffLight::fromToVecExposure). Runs with whichever LM driver the process selected (persistent kernel by default, host loop
    under NALO_TRK_HOST_LM=1)."""
    win = small_window
    c = binding.Context(win.w, win.h, win.K, n_slots=2)
    c.frame_upload(0, win.images[win.W - 1]); c.frame_upload(1, (win.images[win.W] * 1.25 + 3.0).astype(np.float32))
    Ku, Kv, nid, hdi = tracker_inputs(win)
    trk = orc.Tracker(win.w, win.h, win.levels, win.K)
    dI_ref, _ = orc.make_images(win.images[win.W - 1], win.levels)
    dI_new, _ = orc.make_images((win.images[win.W] * 1.25 + 3.0).astype(np.float32), win.levels)
    trk.set_ref(dI_ref, Ku, Kv, nid, hdi)
    c.trk_set_ref(0, Ku, Kv, nid, hdi)
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.85)
    # (i) brightness change a' = 1.25 I + 3 with exposures (1, 1.25): the affine pair is recovered the same way by both
    ok_o, T_o, aff_o, lr_o, _ = trk.track(dI_new, T0, [0, 0], [0, 0], [1.0, 1.25], win.levels - 1)
    ok, T, aff, lr, lf, nev = c.trk_track(1, T0, [0, 0], [0, 0], [1.0, 1.25], win.levels - 1)
    assert ok == ok_o == 1 and pose_dist(T, T_o) < 1e-5 and np.abs(aff - aff_o).max() < 1e-3
    assert abs(aff[0]) + abs(aff[1]) > 1e-2 and pose_dist(T, true_rel_pose(win, win.W - 1, win.W)) < 3e-3   # the affine pair moved, the pose is still recovered
    # (ii) abort: a tiny minResForAbort makes the coarsest level fail
    mr = np.full(5, 1e-3)
    ok_o, T_o, aff_o, lr_o, _ = trk.track(dI_new, T0, [0, 0], [0, 0], [1.0, 1.25], win.levels - 1, min_res=mr)
    ok, T, aff, lr, lf, nev = c.trk_track(1, T0, [0, 0], [0, 0], [1.0, 1.25], win.levels - 1, min_res=mr)
    assert ok == ok_o == 0
    assert np.array_equal(T, np.asarray(T0).reshape(3, 4)) and np.array_equal(aff, [0, 0])      # outputs untouched
    top = win.levels - 1
    assert abs(lr[top] - lr_o[top]) < 1e-4 * lr_o[top] and np.isnan(lr[:top]).all() and np.isnan(lr_o[:top]).all()
    c.close()


_LOST_BLOCK_SCRIPT = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests"); sys.path.insert(0, sys.argv[1] + "/oracle")
import nalo_pkg
nalo_pkg.load()
from helpers import tracker_inputs
from nalo_slam_amd import binding, synth
win = synth.make_window(w=640, h=480, W=4, P=400, seed=7)
c = binding.Context(win.w, win.h, win.K, n_slots=2)
c.frame_upload(0, win.images[win.W - 1]); c.frame_upload(1, win.images[win.W])
Ku, Kv, nid, hdi = tracker_inputs(win)
c.trk_set_ref(0, Ku, Kv, nid, hdi)
T0 = np.asarray(json.loads(sys.argv[2]))
out = []
for _ in range(2):                                   # frame 1: the launch is reported lost and redone from the host; frame 2: host loop straight away
    ok, T, aff, lr, lf, nev = c.trk_track(1, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)
    out.append(dict(ok=int(ok), T=np.asarray(T).tolist(), aff=np.asarray(aff).tolist(), nev=int(nev)))
c.close()
print("RESULT " + json.dumps(out))
"""


def test_lost_workgroup_degrades_to_the_host_driven_loop(ctx, small_window, tmp_path):
    """Hygiene item of the round-1 review: trk_lm_kernel needs its workgroups co-resident. When one never arrives (CUs held by another context) the launch ends
    in its bounded poll and nalo_trk_track REDOES the frame with the host-driven LM loop (the same fused evaluation kernel per step - still the HIP path) and keeps
    to it for the context. NALO_LM_TEST_TIMEOUT makes a context's first launch report the loss; the env is read once per process, hence the child process."""
    import json, os, subprocess, sys
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win)
    ctx.trk_set_ref(win.W - 1, Ku, Kv, nid, hdi)
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.8)
    ok, T, aff, lr, lf, nev = ctx.trk_track(win.W, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "lost_block.py"
    script.write_text(_LOST_BLOCK_SCRIPT)
    env = dict(os.environ, NALO_LM_TEST_TIMEOUT="1", NALO_HOST_TIMING="1")      # the second switch: the host-side wall-clock accounting printed at nalo_destroy
    p = subprocess.run([sys.executable, str(script), root, json.dumps(np.asarray(T0).tolist())], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "drives the tracker's LM loop from the host" in p.stderr
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for r in res:
        assert r["ok"] == ok == 1
        assert pose_dist(np.asarray(r["T"]), T) < 1e-5 and np.abs(np.asarray(r["aff"]) - aff).max() < 1e-3
    assert p.stderr.count("drives the tracker's LM loop from the host") == 1          # latched: the second frame did not try the persistent kernel again
    acct = [l for l in p.stderr.splitlines() if l.startswith("[nalo host]")]
    assert any("trk_track" in l and "calls=     2" in l for l in acct) and any("trk_set_ref" in l for l in acct), p.stderr[-1500:]
    # the third switch of the library: NALO_TRK_HOST_LM=1 selects the host-driven loop from the first frame on (no lost launch, no message), same poses
    env = dict({k: v for k, v in os.environ.items() if k not in ("NALO_LM_TEST_TIMEOUT", "NALO_HOST_TIMING")}, NALO_TRK_HOST_LM="1")
    p = subprocess.run([sys.executable, str(script), root, json.dumps(np.asarray(T0).tolist())], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "drives the tracker's LM loop from the host" not in p.stderr and "[nalo host]" not in p.stderr
    for r in json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]):
        assert r["ok"] == 1 and pose_dist(np.asarray(r["T"]), T) < 1e-5 and np.abs(np.asarray(r["aff"]) - aff).max() < 1e-3


def test_sharded_tracker_sums_and_tracks_like_one_gpu(small_window):
    """SURVEY 8(e), tracker: two contexts hold the whole reference and evaluate one half of every level's point cloud each (nalo_trk_set_shard); the 52 sums of an
    evaluation meet in the all-reduce hook (here: two threads summing through the host, as in tests/test_shard_gpu.py). What the ranks agree on must be what one
    context computes alone: a single evaluation's stats / H / b (linearity: only the fp32 block partials group differently) and the tracked pose (the same LM path;
    the ranks bit-equal among themselves because they read the same summed buffer)."""
    import threading
    from helpers import dev_read_f64, dev_write_f64
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win)
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.8)
    aff = np.array([0.98, 1.5], np.float32)

    def make():
        c = binding.Context(win.w, win.h, win.K, n_slots=2)
        c.frame_upload(0, win.images[win.W - 1]); c.frame_upload(1, win.images[win.W])
        c.trk_set_ref(0, Ku, Kv, nid, hdi)                      # replicated: every rank passes all the reference points
        return c
    one = make()
    one.trk_set_shard(0, 1)                                     # world 1 = off: the single-GPU tracker (persistent LM kernel)
    st1, H1, b1 = one.trk_eval(1, 0, T0, aff, 0.3, 20.0)
    ok1, T1, aff1, lr1, lf1, nev1 = one.trk_track(1, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)
    one.close()

    world = 2
    bar = threading.Barrier(world)
    bufs, out, err, ncalls = [None] * world, [None] * world, [], [0] * world

    def rank_job(r):
        try:
            c = make()

            def hook(ptr, n):                                   # blocking contract: the library drained its stream before the call
                assert n == 52
                bufs[r] = dev_read_f64(ptr, n)
                bar.wait()
                tot = bufs[0] + bufs[1]
                bar.wait()
                dev_write_f64(ptr, tot)
                ncalls[r] += 1
            c.trk_set_shard(r, world, hook)
            st, H, b = c.trk_eval(1, 0, T0, aff, 0.3, 20.0)
            ok, T, a, lr, lf, nev = c.trk_track(1, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)
            out[r] = dict(st=st, H=H, b=b, ok=ok, T=T, aff=a, lr=lr, nev=nev)
            c.close()
        except Exception as ex:                                 # never leave the other rank waiting in the barrier
            err.append(ex)
            bar.abort()
    ts = [threading.Thread(target=rank_job, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not err, err
    a, b = out
    for k in ("st", "H", "b", "T", "aff", "lr"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True), k      # the ranks read the same sums
    assert a["ok"] == b["ok"] == ok1 == 1 and a["nev"] == b["nev"] == nev1 and ncalls[0] == ncalls[1] == 1 + nev1
    assert st1[1] == a["st"][1] and abs(a["st"][0] - st1[0]) < 1e-6 * st1[0]                  # the same residual count, the same energy
    assert rel_err(a["H"], H1) < 1e-6 and rel_err(a["b"], b1) < 1e-6
    # the LM loop amplifies the regrouped fp32 partials (measured 3.7e-7): a fifth of the 1e-5 bar of BASELINE.json is the bound
    d_aff = np.abs(np.asarray(a["aff"]) - aff1)                  # b is in grey levels and scaled by SCALE_B = 1000 inside the LM: measured 8.8e-5
    assert pose_dist(a["T"], T1) < 2e-6 and d_aff[0] < 1e-5 and d_aff[1] < 1e-3


def test_sharded_tracker_stops_when_its_exchange_fails(small_window):
    """the tracker's hook has the contract of nalo_ba_set_allreduce, failure included. A collective that fails (reported through
    nalo_ba_exchange_failed, what the built-in RCCL hooks do) must end the call that issued it and latch the context: a rank that went on with rank-local
    sums would take other LM steps and other evaluation counts than its peers, and the next collectives would mismatch or hang."""
    win = small_window
    Ku, Kv, nid, hdi = tracker_inputs(win)
    T0 = orc.se3_exp(orc.se3_log(true_rel_pose(win, win.W - 1, win.W)) * 0.8)
    c = binding.Context(win.w, win.h, win.K, n_slots=2)
    c.frame_upload(0, win.images[win.W - 1]); c.frame_upload(1, win.images[win.W])
    c.trk_set_ref(0, Ku, Kv, nid, hdi)
    n_calls = [0]

    def hook(ptr, n):                                           # the "sum" of a lone rank is the identity; the third collective reports a failure
        n_calls[0] += 1
        if n_calls[0] == 3:
            c.ba_exchange_failed("link down (tracker test)")
    c.trk_set_shard(0, 2, hook)                                 # rank 0 of 2: the host-driven LM loop with the hook per evaluation
    c.trk_eval(1, 0, T0, np.array([1.0, 0.0], np.float32), 0.0, 20.0)           # collective 1: fine
    with pytest.raises(RuntimeError, match="link down"):
        c.trk_track(1, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)               # collective 2 fine, collective 3 fails: the LM loop stops there
    assert n_calls[0] == 3
    with pytest.raises(RuntimeError, match="cross-rank sum"):
        c.trk_track(1, T0, [0, 0], [0, 0], [1, 1], win.levels - 1)               # latched
    with pytest.raises(RuntimeError, match="cross-rank sum"):
        c.trk_eval(1, 0, T0, np.array([1.0, 0.0], np.float32), 0.0, 20.0)
    assert n_calls[0] == 3                                      # nothing reached the hook any more
    c.close()
