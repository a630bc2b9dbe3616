#!/bin/bash
# HBM-side byte counters of the roofline kernel, one counter per pass (never together with a trace), csv output.
# usage (through gpurun): bash scripts/prof_pmc.sh <tag> <workload>
tag=${1:-pmc}; wl=${2:-stress250k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/$tag/$ctr -o c -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/$tag.$ctr.log 2>&1 || exit 1
  echo "$ctr done"
done
