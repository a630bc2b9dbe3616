"""does the number of (sleeping) host threads in the process change what a blocking hipStreamSynchronize costs? the staged / resident activation calls before and after
64 idle threads exist (bench.py's all-cores CPU baseline leaves such a pool behind)"""
import sys, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import bench
def show(tag, r): print(tag, {k: r[k] for k in ("optimize_call_us", "optimize_staged_call_us", "trace_call_us")}, flush=True)
show("before       ", bench.imm_leg(cpu=False))
ev = threading.Event()
ths = [threading.Thread(target=ev.wait, daemon=True) for _ in range(64)]
for t in ths: t.start()
show("64 idle threads", bench.imm_leg(cpu=False))
ev.set()
for t in ths: t.join()
show("threads gone ", bench.imm_leg(cpu=False))
