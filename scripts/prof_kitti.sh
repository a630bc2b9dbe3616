#!/bin/bash
# kernel-trace profile of the default bench workload; writes gpurun_out/<tag>/*kernel_stats.csv
tag=${1:-prof}
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1
