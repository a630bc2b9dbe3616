#!/bin/bash
# A/B of prebuilt library variants (nalo-slam_amd/variants/*.so) on the headline window, same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
cp nalo-slam_amd/libnalo_gpu.so /tmp/keep.so
for v in "$@"; do
  cp nalo-slam_amd/variants/$v.so nalo-slam_amd/libnalo_gpu.so
  timeout -k 10 200 python bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], (d.get('roofline_kitti00_8kf') or d.get('roofline') or {}).get('avg_us'))" || { echo "$v failed"; cp /tmp/keep.so nalo-slam_amd/libnalo_gpu.so; exit 1; }
done
cp /tmp/keep.so nalo-slam_amd/libnalo_gpu.so
