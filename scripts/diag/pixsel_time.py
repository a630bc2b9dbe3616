"""nalo_pixsel_make_maps on a KITTI-sized frame: per-call wall time over many calls (median / mean), frame_rebuild before each like a new keyframe"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench, orc
from bench import synth, binding
cfg = bench.WORKLOADS["kitti00_8kf"]
win = synth.make_window(w=cfg["w"], h=cfg["h"], W=2, P=64, seed=9, n_extra=0)
rp, draws = orc.pixsel_libc_tables(win.w * win.h)
c = binding.Context(win.w, win.h, win.K, n_slots=1)
c.frame_upload(0, win.images[1]); c.pixsel_set_random(rp, draws)
m, num, pot = c.pixsel_make_maps(0, 1500.0, 3)
for rep in range(3):
    ts = []
    for _ in range(200):
        c.frame_rebuild(0); c.sync()
        t0 = time.perf_counter(); c.pixsel_make_maps(0, 1500.0, pot); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("make_maps: median %.1f us, mean %.1f us, p10 %.1f" % (np.median(ts), ts.mean(), np.percentile(ts, 10)), flush=True)
c.close()
