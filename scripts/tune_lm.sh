#!/bin/bash
# sweeps the persistent LM kernel's grid size / barrier flavour on the GPU box (run through gpurun)
for cfg in "NALO_LM_BLOCKS=1" "NALO_LM_BLOCKS=4" "NALO_LM_BLOCKS=16 NALO_LM_LIGHT=0" "NALO_LM_BLOCKS=16" "NALO_LM_BLOCKS=128 NALO_LM_LIGHT=0" "NALO_LM_BLOCKS=128" "NALO_TRK_HOST_LM=1"; do
  echo "== $cfg"
  env $cfg NALO_HOST_TIMING=1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | grep -oE "\"value\": [0-9.]+|\"trk_lm\": \{[^}]*\}|nalo host\] trk_track.*" || exit 1
done
