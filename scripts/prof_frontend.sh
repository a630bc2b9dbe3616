#!/bin/bash
# rocprofv3 kernel statistics of the front-end legs (pyramid / dense map / ingest / tracker evaluation at 1920x1072). usage: prof_frontend.sh <tag>
tag=${1:-fe}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/fe_stats -o k -- python3 scripts/diag/frontend_prof.py > gpurun_out/$tag/fe_line.json 2> gpurun_out/$tag/fe.err || exit 1
echo "frontend stats done"
