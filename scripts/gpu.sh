#!/bin/bash
# gpurun wrapper: retries ONLY when no box/slot was free (exit code 3: nothing ran, nothing charged); any other result is returned as is.
# usage: scripts/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
