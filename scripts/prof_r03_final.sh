#!/bin/bash
# the round-3 evidence set in one call: default bench line, rocprofv3 kernel statistics (headline, stress250k, front end, initialiser, emulated 8-rank shard) and
# the emulated strong-scaling probe. usage: prof_r03_final.sh <tag>
tag=${1:-r03final}
bash scripts/prof_r03.sh $tag || exit 1
bash scripts/prof_frontend.sh $tag || exit 1
bash scripts/prof_init.sh $tag > gpurun_out/$tag/init_top.txt || exit 1
NALO_BENCH_EMULATE_WORLD=8 bash scripts/prof_any.sh $tag/shard8 scripts/diag/shard_fixed.py 30 > gpurun_out/$tag/shard8_top.txt || exit 1
bash scripts/shard_scaling_probe.sh > gpurun_out/$tag/shard_scaling_probe.log 2>&1 || exit 1
echo "all done"
