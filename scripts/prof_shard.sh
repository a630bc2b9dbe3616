#!/bin/bash
# kernel-trace profile of the sharded-BA leg at P points (default 1M); writes gpurun_out/<tag>/*kernel_stats.csv
tag=${1:-prof_shard}; P=${2:-1000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o k -- python3 scripts/run_shard_leg.py $P > gpurun_out/$tag.log 2>&1
python3 scripts/show_stats.py gpurun_out/$tag 2>/dev/null | head -30
