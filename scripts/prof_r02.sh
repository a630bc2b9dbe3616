#!/bin/bash
# round-2 profile set (run through gpurun): tests, the bench line, rocprofv3 kernel statistics of the default and the stress250k command, PMC passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02
timeout -k 10 400 python bench.py > gpurun_out/r02/default_bench_line.json 2> gpurun_out/r02/default_bench.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/default_stats -o k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02/default_stats_line.json 2> gpurun_out/r02/default_stats.err || exit 1
echo "default stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/stress_stats -o k -- python3 bench.py --workload stress250k --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r02/stress_stats_line.json 2> gpurun_out/r02/stress_stats.err || exit 1
echo "stress stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/r02/pmc/$ctr -o c -- python3 bench.py --workload stress250k --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r02/pmc_$ctr.log 2>&1 || exit 1
  echo "$ctr done"
done
for ctr in TA_BUSY_avr GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum VALUBusy MemUnitStalled SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/r02/pmcx/$ctr -o c -- python3 bench.py --workload stress250k --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r02/pmcx_$ctr.log 2>&1 || { echo "$ctr FAILED"; continue; }
  echo "$ctr done"
done
