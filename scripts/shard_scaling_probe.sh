#!/bin/bash
# per-rank work of the shard1m window at N = 1, 2, 4, 8 emulated on one GPU (1-rank RCCL group, P/N points): upper bound of the strong scaling
for P in 1000000 500000 250000 125000; do
  echo "== P=$P"
  NALO_BENCH_SHARD_P=$P NALO_HOST_TIMING=1 timeout -k 10 300 python -c "
import bench, json, torch
r = bench.shard_leg(0, 1, 0, None, torch, steps=4, warmup=1)
print({k: r[k] for k in ('keyframes_per_s','ms_per_keyframe','points_per_rank')}, r.get('ba_linearize'))
" 2>&1 | grep -E "keyframes_per_s|nalo host\] (ba_optimize|ba_restore|ba.solve.fetch_wait|ba.solve_system)" || exit 1
done
