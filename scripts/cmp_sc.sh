#!/bin/bash
# vector-ALU vs MFMA SYRK in ba_sc_kernel on the three bench window sizes (kernel times from the library's own profile)
for P in 16000 250000 1000000; do
  for cfg in "NALO_SC_VALU=1" "NALO_SC_MFMA=1"; do
    echo "== P=$P $cfg"
    env $cfg timeout -k 10 300 python scripts/run_shard_leg.py $P 2>&1 | grep -oE "'ba_sc_us': [0-9.]+|'ba_reduce_us': [0-9.]+|'ba_linearize_us': [0-9.]+|'keyframes_per_s': [0-9.]+" || exit 1
  done
done
