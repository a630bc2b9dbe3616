"""front-end legs (bench.frontend_legs) as one short table: scope, us, frac. usage: fe_short.py [rounds]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
r = bench.frontend_legs(rounds=int(sys.argv[1]) if len(sys.argv) > 1 else 30)
print("  ".join("%s %.2f us (%.3f)" % (k, v["avg_us"], v["frac"]) for k, v in r.items() if isinstance(v, dict)))
