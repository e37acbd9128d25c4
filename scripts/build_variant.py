#!/usr/bin/env python
"""Build a variant of libcbgx.so into ab_libs/<name>.so with extra compiler flags (timing ablations / A-B experiments; ab_libs/ is
git-ignored but travels to the GPU box).  Usage: python scripts/build_variant.py <name> [-DFLAG ...]
Run on the box with CBGX_LIBRARY=$PWD/ab_libs/<name>.so python bench.py ...   (scripts/ab_fwd.sh loops over all of them)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cbgbench_amd import build as B  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
objdir = os.path.join("/tmp", "cbgx_variant_" + name)
os.makedirs(objdir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "ab_libs"), exist_ok=True)
flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-munsafe-fp-atomics"] + extra
jobs, objs = [], []
for src in B.sources(False):
    obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
    objs.append(obj)
    jobs.append(subprocess.Popen([B.hipcc()] + flags + ["-c", src, "-o", obj], stderr=subprocess.DEVNULL))
for j in jobs:
    assert j.wait() == 0, j.args
out = os.path.join(ROOT, "ab_libs", name + ".so")
subprocess.run([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC"] + objs + ["-o", out], check=True, stderr=subprocess.DEVNULL)
print(out)
