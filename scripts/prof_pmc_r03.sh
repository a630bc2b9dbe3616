#!/bin/bash
# round-3 HBM traffic of ba_linearize: separate --pmc FETCH_SIZE / WRITE_SIZE passes (never with a trace) on stress250k and on the headline window.
# usage: prof_pmc_r03.sh   then, in the container: NALO_TRAFFIC_OUT=profiles/traffic_r03.json python scripts/make_traffic.py r03pmc/stress:stress250k r03pmc/kitti:kitti00_8kf
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03pmc
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/r03pmc/stress/$ctr -o c -- python3 bench.py --workload stress250k --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r03pmc/stress_$ctr.log 2>&1 || exit 1
  echo "stress $ctr done"
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/r03pmc/kitti/$ctr -o c -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r03pmc/kitti_$ctr.log 2>&1 || exit 1
  echo "kitti $ctr done"
done
