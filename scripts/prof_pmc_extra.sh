#!/bin/bash
# extra derived / raw counters of the BA kernels, ONE counter per pass (never together with a trace), csv output
# usage (through gpurun): bash scripts/prof_pmc_extra.sh <tag> <workload> CTR1 CTR2 ...
tag=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for ctr in "$@"; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/$tag/$ctr -o c -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/$tag.$ctr.log 2>&1 || { echo "$ctr FAILED"; continue; }
  echo "$ctr done"
done
