"""setFirst / trackFrame wall clock on the KITTI frame shape (bench.py's init leg without the CPU port); run with NALO_HOST_TIMING=1 for the host-side split."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
for _ in range(2):
    print(json.dumps(bench.init_leg(cpu=False, frames=4)))
