#!/bin/bash
# copies the evidence set of scripts/prof_r04.sh from gpurun_out/<tag>/ into profiles/ under the round's names, builds traffic_r04.json and INDEX.json
# usage (in the container): install_profiles.sh <gpurun tag> [round, default r04]
tag=$1; r=${2:-r04}; src=gpurun_out/$tag; dst=profiles
cpk() { [ -f "$1" ] && cp "$1" "$2"; }
cpk $src/default_bench_line.json $dst/${r}_default_bench_line.json
cpk $src/kitti_stats/k_kernel_stats.csv $dst/${r}_kitti00_kernel_stats.csv; cpk $src/kitti_line.json $dst/${r}_kitti00_bench_line_under_profiler.json; cpk $src/kitti_timeline.txt $dst/${r}_kitti00_timeline.txt
cpk $src/stress_stats/k_kernel_stats.csv $dst/${r}_stress250k_kernel_stats.csv; cpk $src/stress_line.json $dst/${r}_stress250k_bench_line_under_profiler.json; cpk $src/stress_timeline.txt $dst/${r}_stress250k_timeline.txt
cpk $src/fe_stats/k_kernel_stats.csv $dst/${r}_frontend_kernel_stats.csv; cpk $src/fe_line.json $dst/${r}_frontend_line.json
cpk $src/init_stats/k_kernel_stats.csv $dst/${r}_initializer_kernel_stats.csv; cpk $src/init_line.json $dst/${r}_initializer_lines.json
cpk $src/shard8_stats/k_kernel_stats.csv $dst/${r}_shard1m_rank0of8_kernel_stats.csv; cpk $src/shard8_line.txt $dst/${r}_shard1m_rank0of8_line.txt; cpk $src/shard8_timeline.txt $dst/${r}_shard1m_rank0of8_timeline.txt
cpk $src/shard_scaling_probe.log $dst/${r}_shard_scaling_probe.log
cpk $src/pmc_summary.json $dst/${r}_pmc_summary.json
[ -f $dst/${r}_pmc_summary.json ] && python3 scripts/make_traffic.py --install $dst/${r}_pmc_summary.json $dst/traffic_${r}.json > /dev/null
python3 scripts/make_index.py $r > /dev/null && echo "profiles/INDEX.json written"
ls $dst | grep "^${r}_\|traffic_${r}\|INDEX"
