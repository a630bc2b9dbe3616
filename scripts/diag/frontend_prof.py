"""the front-end roofline legs of bench.py alone (pyramid, dense map, ingest, tracker evaluation at 1920x1072): run under rocprofv3 --kernel-trace --stats"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
print(json.dumps(bench.frontend_legs(rounds=int(sys.argv[1]) if len(sys.argv) > 1 else 30)))
