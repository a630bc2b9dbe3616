#!/bin/bash
# persistent LM kernel: workgroup count sweep (NALO_LM_BLOCKS) on the headline window, current library
cd "$GRAFT_REPO_ROOT" || exit 1
for nb in "$@"; do
  NALO_LM_BLOCKS=$nb timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks $nb', d['value'], d['ms_per_step'])" || exit 1
done
