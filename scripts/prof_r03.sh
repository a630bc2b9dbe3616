#!/bin/bash
# round-3 profile set (run through gpurun): bench line + rocprofv3 kernel statistics of the headline-only and the stress250k commands. usage: prof_r03.sh <tag>
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
timeout -k 10 500 python bench.py > gpurun_out/$tag/default_bench_line.json 2> gpurun_out/$tag/default_bench.err || exit 1
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kitti_stats -o k -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/$tag/kitti_stats_line.json 2> gpurun_out/$tag/kitti_stats.err || exit 1
echo "kitti stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stress_stats -o k -- python3 bench.py --workload stress250k --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/$tag/stress_stats_line.json 2> gpurun_out/$tag/stress_stats.err || exit 1
echo "stress stats done"
