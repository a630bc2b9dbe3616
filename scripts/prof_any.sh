#!/bin/bash
# rocprofv3 kernel statistics of any driver script. usage: prof_any.sh <tag> <script.py> [env assignments are inherited]
tag=$1; drv=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o k -- python3 $drv > gpurun_out/$tag/line.log 2> gpurun_out/$tag/err.log || exit 1
f=$(ls gpurun_out/$tag/stats/*kernel_stats.csv | head -1)
cut -d, -f1-4 "$f" | head -${3:-24}
