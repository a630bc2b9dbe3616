"""Builds nalo-slam_amd/variants/<name>.so = the current library with ONE source file recompiled under extra flags (the other objects are reused from
nalo-slam_amd/build). usage: python scripts/mkvariant.py <name> <file.hip> [-DX=1 ...]   (same-box A/B through scripts/ab.sh)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nalo-slam_amd"))
import build as B
name, fname, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()                                                   # the default objects
bdir, vdir = os.path.join(B.HERE, "build"), os.path.join(B.HERE, "variants")
os.makedirs(vdir, exist_ok=True)
objs = []
for s in B.SRC:
    tag = "nc" if s in B.NO_CONTRACT else "fc"
    o = os.path.join(bdir, s + "." + tag + ".o")
    if s == fname:
        o = os.path.join(vdir, name + "." + s + ".o")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(B.HERE, "csrc", s), "-o", o, "-Wno-unused-result"] + flags
        if s in B.NO_CONTRACT: cmd += ["-ffp-contract=off"]
        subprocess.check_call(cmd)
    objs.append(o)
out = os.path.join(vdir, name + ".so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lz"])
print(out)
