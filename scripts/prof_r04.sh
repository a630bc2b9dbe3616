#!/bin/bash
# The round-4 evidence set in one gpurun call. usage: prof_r04.sh <tag> [parts: bench kitti stress frontend init shard probe pmc]   (default: all)
#   bench     python bench.py                                              -> <tag>/default_bench_line.json
#   kitti     rocprofv3 --kernel-trace --stats of the headline command    -> <tag>/kitti_stats (+ timeline)
#   stress    rocprofv3 --kernel-trace --stats of  bench.py --workload stress250k --no-extra  = the loop the `roofline` object is measured over (full steps)
#   frontend  front-end legs at 1920x1072;  init: the two-frame initialiser;  shard: rank 0 of an emulated 8-rank shard1m job (+ timeline);  probe: N = 1, 2, 4, 8
#   pmc       separate --pmc FETCH_SIZE / WRITE_SIZE passes (never with a trace) of the stress, headline and front-end commands
tag=${1:-r04}; shift
parts=${*:-bench kitti stress frontend init shard probe pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
has() { [[ " $parts " == *" $1 "* ]]; }
KITTI="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra"
STRESS="bench.py --workload stress250k --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
if has bench; then timeout -k 10 600 python bench.py > gpurun_out/$tag/default_bench_line.json 2> gpurun_out/$tag/default_bench.err || exit 1; echo "bench done"; fi
if has kitti; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kitti_stats -o k -- python3 $KITTI > gpurun_out/$tag/kitti_line.json 2> gpurun_out/$tag/kitti.err || exit 1
  python3 scripts/trace_timeline.py gpurun_out/$tag/kitti_stats ba_restore > gpurun_out/$tag/kitti_timeline.txt; rm -f gpurun_out/$tag/kitti_stats/*kernel_trace.csv; echo "kitti stats done"
fi
if has stress; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stress_stats -o k -- python3 $STRESS > gpurun_out/$tag/stress_line.json 2> gpurun_out/$tag/stress.err || exit 1
  python3 scripts/trace_timeline.py gpurun_out/$tag/stress_stats ba_restore > gpurun_out/$tag/stress_timeline.txt; rm -f gpurun_out/$tag/stress_stats/*kernel_trace.csv; echo "stress stats done"
fi
if has frontend; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/fe_stats -o k -- python3 scripts/diag/frontend_prof.py > gpurun_out/$tag/fe_line.json 2> gpurun_out/$tag/fe.err || exit 1
  rm -f gpurun_out/$tag/fe_stats/*kernel_trace.csv; echo "frontend stats done"
fi
if has init; then
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/init_stats -o k -- python3 scripts/diag/init_time.py > gpurun_out/$tag/init_line.json 2> gpurun_out/$tag/init.err || exit 1
  rm -f gpurun_out/$tag/init_stats/*kernel_trace.csv; echo "init stats done"
fi
if has shard; then
  NALO_BENCH_EMULATE_WORLD=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/shard8_stats -o k -- python3 scripts/diag/shard_fixed.py > gpurun_out/$tag/shard8_line.txt 2> gpurun_out/$tag/shard8.err || exit 1
  python3 scripts/trace_timeline.py gpurun_out/$tag/shard8_stats ba_resub > gpurun_out/$tag/shard8_timeline.txt; rm -f gpurun_out/$tag/shard8_stats/*kernel_trace.csv; echo "shard8 stats done"
fi
if has probe; then bash scripts/shard_scaling_probe.sh > gpurun_out/$tag/shard_scaling_probe.log 2>&1 || exit 1; echo "probe done"; fi
if has pmc; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/$tag/pmc/stress/$ctr -o c -- python3 bench.py --workload stress250k --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/$tag/pmc_stress_$ctr.log 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/$tag/pmc/kitti/$ctr -o c -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/$tag/pmc_kitti_$ctr.log 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/$tag/pmc/fe/$ctr -o c -- python3 scripts/diag/frontend_prof.py > gpurun_out/$tag/pmc_fe_$ctr.log 2>&1 || exit 1
    echo "pmc $ctr done"
  done
  python3 scripts/make_traffic.py --summarise gpurun_out/$tag/pmc > gpurun_out/$tag/pmc_summary.json; rm -rf gpurun_out/$tag/pmc/*/FETCH_SIZE gpurun_out/$tag/pmc/*/WRITE_SIZE
fi
echo "all done"
