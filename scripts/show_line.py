"""one-screen summary of a bench.py JSON line (the last line of the file that starts with '{')"""
import json, sys
txt = [l for l in open(sys.argv[1]) if l.startswith('{')]
d = json.loads(txt[-1])
print("value", d["value"], "ms", d["ms_per_step"], "| first bytes:", open(sys.argv[1]).read(24).replace("\n", " "))
r = d["roofline"]; print("roofline", {k: r.get(k) for k in ("workload", "achieved", "frac", "avg_us", "launches", "traffic", "measured_over")}); print("  stats", r.get("stats")); print("  ba-only loop", r.get("stats_ba_only_loop"))
if "roofline_headline" in d: print("headline roofline", {k: d["roofline_headline"][k] for k in ("frac", "avg_us", "share_of_step")})
s = d.get("stress250k")
if s: print("stress", {k: s[k] for k in s if k not in ("ba_linearize", "traffic", "note")})
if "shard1m" in d: print("shard1m", {k: d["shard1m"].get(k) for k in ("keyframes_per_s", "ms_per_keyframe", "rccl_ranks", "ba_sc_us", "ba_reduce_us", "ba_resub_us")}, d["shard1m"].get("ba_linearize"))
if "pipelined" in d: print("pipelined", d["pipelined"].get("value"), "uploads", d.get("value_with_uploads"), d.get("with_raw_frame_uploads", {}).get("value"))
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "pose", d.get("pose_delta_vs_oracle"))
for k, v in (d.get("frontend_rooflines") or {}).items():
    if isinstance(v, dict): print("  fe", k, v.get("avg_us"), v.get("frac"))
if "immature" in d: print("imm", {k: d["immature"].get(k) for k in ("trace_kernel_us", "trace_call_us", "trace_staged_call_us", "trace_resident_us_per_frame", "optimize_kernel_us", "optimize_call_us")})
if "pixel_selector" in d: print("pixsel", d["pixel_selector"])
if "initializer" in d: print("init", d["initializer"])
print("launcher", d.get("launcher"), "rccl_ranks", d.get("rccl_ranks"))
