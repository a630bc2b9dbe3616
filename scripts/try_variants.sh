#!/bin/bash
# times prebuilt library variants (nalo-slam_amd/variants/*.so, built in the container with different NALO_CXXFLAGS) on the stress and the shard window
cd "$GRAFT_REPO_ROOT" || exit 1
cp nalo-slam_amd/libnalo_gpu.so /tmp/keep.so
for v in "$@"; do
  cp nalo-slam_amd/variants/$v.so nalo-slam_amd/libnalo_gpu.so
  for wl in stress250k shard1m; do
    timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$v $wl', d['value'], r.get('avg_us'), r.get('frac'))" || { echo "$v $wl failed"; cp /tmp/keep.so nalo-slam_amd/libnalo_gpu.so; exit 1; }
  done
done
cp /tmp/keep.so nalo-slam_amd/libnalo_gpu.so
