#!/bin/bash
# rocprofv3 kernel statistics of the two-frame initialiser (setFirst + trackFrame on the KITTI frame shape). usage: prof_init.sh <tag>
tag=${1:-init}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/init_stats -o k -- python3 scripts/diag/init_time.py > gpurun_out/$tag/init_line.json 2> gpurun_out/$tag/init.err || exit 1
f=$(ls gpurun_out/$tag/init_stats/*kernel_stats.csv | head -1)
cut -d, -f1-4 "$f" | head -16
