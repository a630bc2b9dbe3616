"""Per-launch HBM traffic of the kernels from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/prof_r04.sh pmc; never collected with a trace).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B-per-lane loads at 64 B (MI355X_MICROARCH.md, HBM section), so the
read side is doubled: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
  make_traffic.py --summarise <dir>     on the GPU box: <dir>/<run>/<CTR>/**/*counter_collection.csv -> one JSON {run: {kernel: {FETCH_SIZE_KiB, WRITE_SIZE_KiB, launches}}} on stdout
  make_traffic.py --install <summary.json> <out.json>   in the container: the summary -> profiles/traffic_rNN.json in the form bench.py's load_traffic reads"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    n = re.sub(r"^void ", "", name).replace("nalo::", "")
    return n.split("(")[0]


def summarise(root):
    out = {}
    for run in sorted(os.listdir(root)):
        per = defaultdict(lambda: defaultdict(list))
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(root, run, ctr, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr:
                        per[short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
        out[run] = {k: {"FETCH_SIZE_KiB": round(sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1), 1), "WRITE_SIZE_KiB": round(sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1), 1),
                        "launches": [len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"])]} for k, v in per.items() if v["FETCH_SIZE"] and v["WRITE_SIZE"]}
    return out


WANT = {   # traffic key (bench.py load_traffic) -> (run, kernel-name prefix)
    "stress250k": ("stress", "ba_linearize_kernel<0, 0, 256>"), "stress250k_ba_sc": ("stress", "ba_sc_kernel<4, 1"), "stress250k_ba_resub": ("stress", "ba_resub_kernel<true"),
    "stress250k_ba_reduce": ("stress", "ba_reduce_kernel"), "stress250k_ba_stitch": ("stress", "ba_stitch_kernel"), "stress250k_trk_lm": ("stress", "trk_lm_kernel"),
    "kitti00_8kf": ("kitti", "ba_linearize_kernel<0, 0, 64>"), "kitti00_8kf_trk_lm": ("kitti", "trk_lm_kernel"), "kitti00_8kf_ba_sc": ("kitti", "ba_sc_kernel<4, 4"),
    "fe_pyramid": ("fe", "pyr_one_pass_kernel"), "fe_pyramid_tail": ("fe", "pyr_grad_tail_kernel"), "fe_ingest": ("fe", "ingest_kernel"), "fe_dense_rows": ("fe", "dense_rows_kernel"),
    "fe_dense_count": ("fe", "dense_count_kernel"), "fe_dense_write": ("fe", "dense_write_kernel"), "fe_trk_eval": ("fe", "trk_eval_kernel"),
}


def install(summary_path, out_path):
    S = json.load(open(summary_path))
    out = {}
    for key, (run, prefix) in WANT.items():
        hit = [(k, v) for k, v in S.get(run, {}).items() if k.startswith(prefix)]
        if not hit:
            continue
        k, v = max(hit, key=lambda kv: kv[1]["launches"][0])
        out[key] = int((2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024)
        out[key + "_raw"] = dict(v, kernel=k, note="per launch; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests as 64 B")
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.endswith("_raw")}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--summarise":
        print(json.dumps(summarise(sys.argv[2]), indent=1))
    else:
        install(sys.argv[2], sys.argv[3])
