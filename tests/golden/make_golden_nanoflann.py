"""Regenerates tests/golden/golden_nanoflann_r02.npz — the ONE fixture of this repository produced by the REFERENCE's own code: the reference's header-only
k-d tree (src/util/nanoflann.h), compiled from where it lies into oracle/_ref/libref_nanoflann.so behind our driver (oracle/ref_nanoflann.cpp), answers the
two queries of CoarseInitializer::makeNN (FullSystem/CoarseInitializer.cpp:992-1069) on the point sets CoarseInitializer::setFirst selects on a seeded synthetic
frame: the 10 nearest neighbours of every point inside its level and the nearest point of (u/2 - 0.25, v/2 - 0.25) one level up. Inputs (point coordinates per
level) and expected outputs (indices incl. the order of equidistant neighbours, fp32 squared distances) are stored; no reference text is.
Needs /root/reference (this container). Run:  python tests/golden/make_golden_nanoflann.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nalo_pkg  # noqa: E402

nalo_pkg.load()
import orc  # noqa: E402
from nalo_slam_amd import synth  # noqa: E402

FRAME = dict(w=320, h=240, W=2, P=20, seed=3, n_extra=0)


def main():
    assert orc.ref_nanoflann() is not None, "oracle/_ref/libref_nanoflann.so missing: make -C oracle (needs /root/reference)"
    win = synth.make_window(**FRAME)
    rp, _ = orc.pixsel_libc_tables(win.w * win.h)
    ini = orc.Initializer(win.w, win.h, win.levels, win.K)      # only used to SELECT the points (coordinates are inputs of the fixture)
    ini.set_first(win.images[0], rp)
    out = {"levels": np.int32(win.levels)}
    for l in range(win.levels):
        u, v = ini.get(l, "u"), ini.get(l, "v")
        out["u%d" % l], out["v%d" % l] = u, v
        idx, dist = orc.ref_nanoflann_knn(u, v, u, v, 10)
        out["nn_idx%d" % l], out["nn_dist%d" % l] = idx, dist
        if l + 1 < win.levels:
            u2, v2 = ini.get(l + 1, "u"), ini.get(l + 1, "v")
            pi, pd = orc.ref_nanoflann_knn(u2, v2, u * np.float32(0.5) - np.float32(0.25), v * np.float32(0.5) - np.float32(0.25), 1)
            out["parent_idx%d" % l], out["parent_dist%d" % l] = pi[:, 0], pd[:, 0]
    np.savez_compressed(os.path.join(HERE, "golden_nanoflann_r02.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
