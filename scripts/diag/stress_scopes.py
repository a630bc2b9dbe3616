"""stress250k BA-only keyframes: keyframes/s and the HIP-event averages of the BA scopes (A/B of kernel variants)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
r = bench.stress_leg(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 6, warmup=2)
print(r["keyframes_per_s"], {k: (v["avg_us"] if isinstance(v, dict) else v) for k, v in r.items() if k in ("ba_linearize", "ba_sc", "ba_resub") or k.endswith("_us")})
