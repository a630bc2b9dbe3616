#!/bin/bash
# per-rank work of the shard1m window at N = 1, 2, 4, 8 emulated on one GPU (1-rank RCCL group, rank 0's spatial share of an N-rank job):
# upper bound of the strong scaling before the real all-reduce latency
for N in 1 2 4 8; do
  echo "== N=$N"
  NALO_BENCH_EMULATE_WORLD=$N NALO_HOST_TIMING=1 timeout -k 10 300 python -c "
import bench, json, torch, gc
gc.disable()          # as bench.py's main does: a generation-2 collection of the harness is a 38 ms stall inside a ten-keyframe region
r = bench.shard_leg(0, 1, 0, None, torch, steps=10, warmup=2)
print({k: r[k] for k in ('keyframes_per_s','ms_per_keyframe','points_per_rank')}, r.get('ba_linearize'), r.get('ba_sc_us'))
" 2>&1 | grep -E "keyframes_per_s|nalo host\] (ba.solve.fetch_wait|ba.solve.host_math)" || exit 1
done
