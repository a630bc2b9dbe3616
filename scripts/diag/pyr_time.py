"""pyramid (makeImages) timing at 1920x1072 / 5 levels and 1224x368 / 4 levels: HIP-event average per frame_rebuild"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import nalo_pkg
nalo_pkg.load()
from nalo_slam_amd import binding
for (w, h) in ((1920, 1072), (1224, 368)):
    K = (0.52 * w, 0.52 * w, (w - 1) / 2.0, (h - 1) / 2.0)
    c = binding.Context(w, h, K, n_slots=1)
    img = (np.random.RandomState(0).rand(h, w) * 255).astype(np.float32)
    c.frame_upload(0, img)
    c.frame_rebuild(0)
    c.profile_select("pyramid"); c.profile_enable(True); c.profile_reset()
    for _ in range(100):
        c.frame_rebuild(0)
    ms, n = c.profile_get("pyramid")
    L = c.levels
    alg = 4.0 * w * h + 16.0 * w * h * sum(0.25 ** l for l in range(L))
    us = ms / n * 1e3
    print("%dx%d L=%d: %.2f us  -> %.3f of 8 TB/s" % (w, h, L, us, alg / (us * 1e-6) / 8e12))
    c.close()
