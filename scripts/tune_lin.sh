#!/bin/bash
# builds ba_linearize variants on the GPU box and times the stress window with each (tuning aid): waves per SIMD x pattern pixels per gather batch
set -e
for cfg in "3 3" "4 3" "4 2" "5 2"; do
  set -- $cfg
  NALO_CXXFLAGS="-DNALO_LIN_COOP_WAVES=$1 -DNALO_LIN_COOP_NPB=$2" python nalo-slam_amd/build.py --force > /dev/null
  echo "== NALO_LIN_COOP_WAVES=$1 NALO_LIN_COOP_NPB=$2"
  for wl in stress250k shard1m; do
  timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$wl', d['value'], r.get('avg_us'), r.get('frac'), {k:round(v['total_ms']/max(v['launches'],1)*1e3,1) for k,v in d.get('kernel_ms',{}).items()})" || echo "$wl failed"
  done
done
python nalo-slam_amd/build.py --force > /dev/null
