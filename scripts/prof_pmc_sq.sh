#!/bin/bash
# one SQ counter pass (<= 8 counters) of a driver script with a prebuilt library variant; prints the per-kernel means of the kernels matching <pattern>
# usage (through gpurun): scripts/prof_pmc_sq.sh <tag> <variant> <driver.py> <pattern> CTR...
tag=$1; var=$2; drv=$3; pat=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cp nalo-slam_amd/libnalo_gpu.so /tmp/keep_pmc.so
cp nalo-slam_amd/variants/$var.so nalo-slam_amd/libnalo_gpu.so
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$tag -o c -- python3 $drv > gpurun_out/$tag.log 2>&1
rc=$?
cp /tmp/keep_pmc.so nalo-slam_amd/libnalo_gpu.so
[ $rc -ne 0 ] && { echo "rocprofv3 failed ($rc)"; tail -5 gpurun_out/$tag.log; exit 1; }
python3 - "$tag" "$pat" <<'PY'
import csv, glob, sys, collections
tag, pat = sys.argv[1], sys.argv[2]
f = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if pat in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:40], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print("%-42s %-28s %14.1f  (n=%d)" % (kn, cn, v / n, n))
PY
