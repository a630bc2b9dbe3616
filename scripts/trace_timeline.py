"""Timeline view of a rocprofv3 --kernel-trace CSV: the kernels between two launches of an anchor kernel, with start offset, duration, gap to the previous
kernel of the same queue. usage: trace_timeline.py <dir-or-csv> [anchor-substring] [which occurrence, default: the middle one] [how many anchors to span, default 1]
Also prints, per kernel name, the duration by POSITION inside the anchor period (e.g. the eight ba_linearize launches of a keyframe)."""
import csv
import glob
import sys
from collections import defaultdict

src = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "ba_restore"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
span = int(sys.argv[4]) if len(sys.argv) > 4 else 1
f = src if src.endswith(".csv") else sorted(glob.glob(src + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ks = [dict(name=r["Kernel_Name"], q=r.get("Queue_Id", "0"), t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"])) for r in rows]
ks.sort(key=lambda k: k["t0"])
idx = [i for i, k in enumerate(ks) if anchor in k["name"]]
if len(idx) < 2:
    print("anchor '%s' seen %d times" % (anchor, len(idx))); sys.exit(1)
w = len(idx) // 2 if which < 0 else which
a, b = idx[w], idx[min(w + span, len(idx) - 1)]
base = ks[a]["t0"]
last_end = {}
print("# %s: occurrence %d of %d of '%s', %d kernels, %.1f us" % (f, w, len(idx), anchor, b - a, (ks[b]["t0"] - base) / 1e3))
busy = 0
for k in ks[a:b]:
    gap = (k["t0"] - last_end[k["q"]]) / 1e3 if k["q"] in last_end else 0.0
    last_end[k["q"]] = k["t1"]
    busy += k["t1"] - k["t0"]
    nm = k["name"].replace("nalo::", "").split("(")[0][:60]
    print("%9.1f  q%-3s %7.1f us  gap %6.1f  %s" % ((k["t0"] - base) / 1e3, k["q"], (k["t1"] - k["t0"]) / 1e3, gap, nm))
print("# kernel time inside the period: %.1f us (sum over queues)" % (busy / 1e3))
# duration by position within the period, over all periods
pos = defaultdict(lambda: defaultdict(list))
for j in range(len(idx) - 1):
    cnt = defaultdict(int)
    for k in ks[idx[j]:idx[j + 1]]:
        nm = k["name"].replace("nalo::", "").split("(")[0][:60]
        pos[nm][cnt[nm]].append((k["t1"] - k["t0"]) / 1e3)
        cnt[nm] += 1
print("# mean duration (us) by position inside the period, over %d periods" % (len(idx) - 1))
for nm, d in sorted(pos.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:14]:
    n = max(d) + 1
    if n > 24:
        continue
    print("%-60s %s" % (nm, " ".join("%.1f" % (sum(d[i]) / len(d[i])) for i in range(n))))
