"""Per-launch HBM traffic of ba_linearize from the rocprofv3 --pmc passes of scripts/prof_pmc.sh.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B-per-lane loads at 64 B
(MI355X_MICROARCH.md, HBM section), so the read side is doubled. Writes profiles/traffic_r02.json (read by bench.py; NALO_TRAFFIC_OUT overrides the path)."""
import csv, glob, json, os, sys

def mean_counter(tag, ctr, kernel):
    vals = []
    for f in glob.glob(f"gpurun_out/{tag}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and (kernel != "ba_linearize" or "<0, 0>" in r["Kernel_Name"] or "<0, 0, " in r["Kernel_Name"]) and r["Counter_Name"] == ctr:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)

out_path = os.environ.get("NALO_TRAFFIC_OUT", "profiles/traffic_r02.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for tag, wl, kern in [(a.split(":") + ["ba_linearize"])[:3] for a in sys.argv[1:]]:      # tag:workload[:kernel] - another kernel's entry is keyed workload_kernel
    f, nf = mean_counter(tag, "FETCH_SIZE", kern)
    w, nw = mean_counter(tag, "WRITE_SIZE", kern)
    if kern != "ba_linearize":
        wl = wl + "_" + kern
    if f is None or w is None:
        print("no counters for", tag); continue
    out[wl] = int((2 * f + w) * 1024)
    out[wl + "_raw"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "launches": [nf, nw],
                        "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests as 64 B"}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
