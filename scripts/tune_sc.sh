#!/bin/bash
# SC work distribution sweep on the N=8-sized shard (125 k points, W=12) and the full 1M window
for cfg in "NALO_SC_SPLIT=1 NALO_SC_BPW=2" "NALO_SC_SPLIT=1 NALO_SC_BPW=1" "NALO_SC_SPLIT=2" "NALO_SC_SPLIT=4"; do
  echo "== 125k $cfg"
  env $cfg NALO_HOST_TIMING=1 timeout -k 10 300 python scripts/run_shard_leg.py 125000 2>&1 | grep -oE "nalo host\] ba.solve.fetch_wait.*|'ba_sc_us': [0-9.]+|'ba_reduce_us': [0-9.]+|'keyframes_per_s': [0-9.]+" || exit 1
done
for cfg in "NALO_SC_BPW=8" "NALO_SC_BPW=4" "NALO_SC_BPW=2" "NALO_SC_BPW=1"; do
  echo "== 1M $cfg"
  env $cfg NALO_HOST_TIMING=1 timeout -k 10 300 python scripts/run_shard_leg.py 1000000 2>&1 | grep -oE "nalo host\] ba.solve.fetch_wait.*|'keyframes_per_s': [0-9.]+" || exit 1
done
