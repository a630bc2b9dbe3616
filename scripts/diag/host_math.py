"""host-side split of the headline window's solve (assemble / LDL^T / the rest) with NALO_HOST_TIMING=1: A/B of host-code variants"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
env = dict(os.environ, NALO_HOST_TIMING="1")
p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "20", "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True)
lines = [l for l in p.stderr.splitlines() if "ba.solve" in l or "ba_optimize" in l]
print("\n".join(l[12:100] for l in lines[-6:]))
import json
print("value", json.loads(p.stdout.strip().splitlines()[-1])["value"])
