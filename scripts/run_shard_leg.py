import os, sys
sys.path.insert(0, os.getcwd())
os.environ["NALO_BENCH_SHARD_P"] = sys.argv[1] if len(sys.argv) > 1 else "125000"
import bench, torch
r = bench.shard_leg(0, 1, 0, None, torch, steps=4, warmup=1)
print(r)
