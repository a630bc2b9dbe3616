"""the activation calls before / after bench.py's CPU baseline legs ran in the same process (what changes the cost of a blocking completion wait?)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); os.chdir(ROOT)
import bench
def show(tag, r): print(tag, {k: r[k] for k in ("optimize_call_us", "optimize_staged_call_us", "trace_call_us")}, flush=True)
win, st6, trk = bench.make_inputs("kitti00_8kf")
show("before           ", bench.imm_leg(cpu=False))
r = bench.cpu_baseline(win, st6, trk, budget_s=5.0, track=True); print("6 threads", round(r["value"], 2))
show("after 6 threads  ", bench.imm_leg(cpu=False))
info = bench.cpu_info(); nall = max(1, min(info["usable_cpus"] or 1, info["physical_cores"] or info["usable_cpus"] or 1, 64))
r = bench.cpu_baseline(win, st6, trk, budget_s=5.0, track=True, nthreads=nall, linearize_mt=True); print(nall, "threads", round(r["value"], 2))
show("after all cores  ", bench.imm_leg(cpu=False))
time.sleep(3)
show("3 s later        ", bench.imm_leg(cpu=False))
