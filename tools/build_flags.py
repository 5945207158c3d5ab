#!/usr/bin/env python3
"""Whole-library A/B builds with extra compiler switches (NOT shipped, NOT loaded by default):
    python tools/build_flags.py TAG FLAG [FLAG ...]     ->  smap_amd/csrc/obj/libsmap_hip_flags_<TAG>.so   (select with SMAP_HIP_LIB=<path>)
e.g.  python tools/build_flags.py maxilp -mllvm -amdgpu-sched-strategy=max-ilp
Every source is compiled with smap_amd/build.py's switches + the given ones (device code scheduling experiments: EXPERIMENTS R6.10)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smap_amd import build as B  # noqa: E402

tag, flags = sys.argv[1], sys.argv[2:]
os.makedirs(B.OBJ, exist_ok=True)


def one(item):
    src, extra = item
    op = os.path.join(B.OBJ, f"flags_{tag}_{src.rsplit('.', 1)[0]}.o")
    subprocess.check_call([B._hipcc()] + B.COMMON + extra + flags + ["-c", os.path.join(B.CSRC, src), "-o", op])
    return op


with ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(one, B.SOURCES))
out = os.path.join(B.OBJ, f"libsmap_hip_flags_{tag}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
