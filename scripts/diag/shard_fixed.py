"""the fixed part of a sharded iteration: rank 0's share of an 8-rank shard1m job emulated on one GPU (NALO_BENCH_EMULATE_WORLD=8), all scopes + host timers
(run with NALO_HOST_TIMING=1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("NALO_BENCH_EMULATE_WORLD", "8")
import torch
import bench
r = bench.shard_leg(0, 1, 0, None, torch, steps=4, warmup=1)
print({k: r[k] for k in r if k not in ("exchange", "collectives_per_linearize")})
