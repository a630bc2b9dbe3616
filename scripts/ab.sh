#!/bin/bash
# A/B of prebuilt library variants (nalo-slam_amd/variants/*.so) with an arbitrary driver script, same box, alternating. usage: ab.sh <script.py> v1 v2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
drv=$1; shift
cp nalo-slam_amd/libnalo_gpu.so /tmp/keep.so
for v in "$@"; do
  cp nalo-slam_amd/variants/$v.so nalo-slam_amd/libnalo_gpu.so
  echo "== $v"
  timeout -k 10 300 python $drv 2>/dev/null || { echo "$v failed"; cp /tmp/keep.so nalo-slam_amd/libnalo_gpu.so; exit 1; }
done
cp /tmp/keep.so nalo-slam_amd/libnalo_gpu.so
