"""profiles/INDEX.json: every roofline figure of the committed bench lines next to the rocprofv3 row it can be recomputed from.
usage: make_index.py <round tag, e.g. r04>     (reads profiles/<tag>_*.json / *.csv, writes profiles/INDEX.json)"""
import csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
PEAK = 8000.0


def line(name):
    p = os.path.join(P, name)
    if not os.path.exists(p):
        return None
    txt = [l for l in open(p) if l.startswith("{")]
    return json.loads(txt[-1]) if txt else None


def rows(name):
    p = os.path.join(P, name)
    return {r["Name"]: r for r in csv.DictReader(open(p))} if os.path.exists(p) else {}


def find(rws, prefix):
    hit = [(k, v) for k, v in rws.items() if k.replace("void ", "").replace("nalo::", "").startswith(prefix)]
    return max(hit, key=lambda kv: int(kv[1]["Calls"])) if hit else (None, None)


def entry(figure, line_file, in_line, csv_file, prefix, alg_bytes, note=""):
    k, r = find(rows(csv_file), prefix)
    e = dict(figure=figure, line_file=line_file, value_in_line=in_line, csv=csv_file, row=k, note=note)
    if r:
        avg_us = float(r["AverageNs"]) / 1e3
        e.update(calls=int(r["Calls"]), avg_us=round(avg_us, 2), min_us=round(float(r["MinNs"]) / 1e3, 2), max_us=round(float(r["MaxNs"]) / 1e3, 2))
        if alg_bytes:
            e.update(alg_bytes=int(alg_bytes), recomputed_GBs=round(alg_bytes / avg_us / 1e3, 1), recomputed_frac=round(alg_bytes / avg_us / 1e3 / PEAK, 4),
                     formula="alg_bytes / (AverageNs of the row) / 8000 GB/s")
    return e


out = {"round": tag, "peak_GBs": PEAK, "how": "each entry: the figure as the bench line prints it (HIP events inside bench.py) and the rocprofv3 --kernel-trace --stats row of the SAME "
       "command (scripts/prof_%s.sh) it can be recomputed from; under the profiler a dispatch is 1-3 %% slower, launch gaps grow" % tag, "figures": []}
F = out["figures"]
d = line("%s_default_bench_line.json" % tag)
s = line("%s_stress250k_bench_line_under_profiler.json" % tag)
k = line("%s_kitti00_bench_line_under_profiler.json" % tag)
fe = line("%s_frontend_line.json" % tag)
if d:
    r = d["roofline"]
    F.append(entry("roofline (ba_linearize on stress250k, full steps) as printed by `python bench.py`", "%s_default_bench_line.json" % tag, dict(avg_us=r.get("avg_us"), frac=r.get("frac"), stats=r.get("stats")),
                   "%s_stress250k_kernel_stats.csv" % tag, "ba_linearize_kernel<0, 0, 256>", r.get("alg_bytes"),
                   "the CSV is the trace of `bench.py --workload stress250k --steps 10 --warmup 2 --no-cpu-baseline --no-extra`: the same full steps (tracking + setCoarseTrackingRef + optimize) as the stress leg of the default run"))
    rh = d.get("roofline_headline")
    if rh:
        F.append(entry("roofline_headline (trk_lm on the KITTI window)", "%s_default_bench_line.json" % tag, dict(avg_us=rh.get("avg_us"), frac=rh.get("frac")), "%s_kitti00_kernel_stats.csv" % tag, "trk_lm_kernel", rh.get("alg_bytes")))
    rk = d.get("roofline_kitti00_8kf")
    if rk:
        F.append(entry("roofline_kitti00_8kf (ba_linearize on the KITTI window: latency bound)", "%s_default_bench_line.json" % tag, dict(avg_us=rk.get("avg_us"), frac=rk.get("frac")), "%s_kitti00_kernel_stats.csv" % tag, "ba_linearize_kernel<0, 0, 64>", rk.get("alg_bytes")))
    st = d.get("stress250k") or {}
    for scope, prefix in (("ba_sc", "ba_sc_kernel<4, 1"), ("ba_resub", "ba_resub_kernel<true")):
        if scope in st:
            F.append(entry("stress250k.%s" % scope, "%s_default_bench_line.json" % tag, dict(avg_us=st[scope].get("avg_us"), frac=st[scope].get("frac")), "%s_stress250k_kernel_stats.csv" % tag, prefix, st[scope].get("alg_bytes"),
                           "the bench scope brackets the launch with recorded events (a few us of pipeline on top of the kernel)"))
    fr = d.get("frontend_rooflines") or {}
    for key, prefix in (("pyramid", "pyr_one_pass_kernel"), ("ingest", "ingest_kernel"), ("dense_map", "dense_write_kernel"), ("trk_eval_250k", "trk_eval_kernel")):
        if key in fr and isinstance(fr[key], dict):
            F.append(entry("frontend_rooflines.%s" % key, "%s_default_bench_line.json" % tag, dict(avg_us=fr[key].get("avg_us"), frac=fr[key].get("frac")), "%s_frontend_kernel_stats.csv" % tag, prefix, None,
                           "a leg of several launches: the row is its largest kernel (pyramid: + pyr_grad_tail_kernel; dense map: dense_rows / dense_count / dense_write)"))
if s:
    r = s["roofline"]
    F.append(entry("roofline of the profiled stress250k command itself (its own line, under the profiler)", "%s_stress250k_bench_line_under_profiler.json" % tag, dict(avg_us=r.get("avg_us"), frac=r.get("frac"), stats=r.get("stats")),
                   "%s_stress250k_kernel_stats.csv" % tag, "ba_linearize_kernel<0, 0, 256>", r.get("alg_bytes"), "HIP events of the timed region vs the profiler's dispatch timestamps over the whole process (warm-up and profile passes included)"))
tr = os.path.join(P, "traffic_%s.json" % tag)
if os.path.exists(tr):
    T = json.load(open(tr))
    out["traffic"] = {"file": "traffic_%s.json" % tag, "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 correction of the guide)",
                      "per_launch_bytes": {k2: v for k2, v in T.items() if not k2.endswith("_raw")}}
json.dump(out, open(os.path.join(P, "INDEX.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
