#!/bin/bash
# builds ba_linearize variants on the GPU box and times the stress window with each (round-1 tuning aid)
set -e
for w in 2 3 4; do
  NALO_CXXFLAGS="-DNALO_LIN_WAVES=$w" python nalo-slam_amd/build.py --force > /dev/null
  echo "== NALO_LIN_WAVES=$w"
  python bench.py --workload stress250k --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_us'], d['roofline']['frac'], {k:round(v['total_ms']/max(v['launches'],1)*1e3,1) for k,v in d['kernel_ms'].items()})"
done
