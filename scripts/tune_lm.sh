#!/bin/bash
# sweeps the persistent LM kernel's grid size on the GPU box (run through gpurun); last line = host-driven LM for comparison
for cfg in "NALO_TRK_DEVICE_LM=1 NALO_LM_BLOCKS=8" "NALO_TRK_DEVICE_LM=1 NALO_LM_BLOCKS=16" "NALO_TRK_DEVICE_LM=1 NALO_LM_BLOCKS=32" "NALO_TRK_DEVICE_LM=1 NALO_LM_BLOCKS=64" "NALO_X=1"; do
  echo "== $cfg"
  env $cfg NALO_HOST_TIMING=1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | grep -oE "\"value\": [0-9.]+|\"fine_track_rmse\": [0-9.]+|\"trk_lm\": \{[^}]*\}|nalo host\] trk_track.*" || exit 1
done
