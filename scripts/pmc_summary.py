"""Per-kernel means of the single-counter rocprofv3 --pmc passes of scripts/prof_r02.sh -> one CSV (counter, kernel, mean per launch, launches).
usage: python scripts/pmc_summary.py gpurun_out/r02 profiles/r02_stress250k_pmc_v2.csv"""
import csv, glob, os, re, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]


def short(name):
    m = re.match(r"(?:void )?nalo::(\w+?)(?:_kernel)?(<[^>]*>)?\(", name)
    if not m:
        return name.split("(")[0]
    return m.group(1) + (m.group(2).replace(" ", "") if m.group(2) and m.group(1) == "ba_linearize" else "")


rows = []
for d in sorted(glob.glob(os.path.join(root, "pmc", "*")) + glob.glob(os.path.join(root, "pmcx", "*"))):
    ctr = os.path.basename(d)
    acc = defaultdict(lambda: defaultdict(float))                 # kernel -> dispatch id -> sum over the counter's instances (XCDs / SEs)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                acc[short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, disp in sorted(acc.items()):
        if k.startswith(("ba_", "trk_", "pyr_")):
            rows.append((ctr, k, sum(disp.values()) / len(disp), len(disp)))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "mean_per_launch", "launches"])
    for r in rows:
        w.writerow([r[0], r[1], "%.1f" % r[2], r[3]])
print(open(out).read()[:200], len(rows), "rows")
