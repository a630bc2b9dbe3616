#!/bin/bash
# times the kitti00 keyframe with several shapes of the device-resident LM kernel (round-1 tuning aid)
set -e
for cfg in "1024 2" "1024 1" "512 4" "512 2" "256 4"; do
  set -- $cfg
  NALO_CXXFLAGS="-DNALO_LM_THREADS=$1 -DNALO_LM_G=$2" python nalo-slam_amd/build.py --force > /dev/null
  echo "== threads=$1 G=$2"
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['total_ms']/max(v['launches'],1)*1e3,1) for k,v in d['kernel_ms'].items() if v['launches']})"
done
