#!/usr/bin/env python3
"""Ablated builds of the persistent conv kernel for bottleneck experiments (NOT shipped, never loaded by default):
smap_amd/csrc/obj/libsmap_hip_pabl<N>.so = the product objects + csrc/convp.hip with -DSMAP_CONVP_ABLATE=N, selected by
SMAP_HIP_LIB=<path>.  N bits: 1 no LDS-DMA (the loaders keep their barriers), 2 no ds_read / MFMA, 4 no global stores,
8 no epilogue.      python tools/build_convp_variants.py 1 2 4 8 3"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smap_amd import build as B  # noqa: E402

B.build_lib()
extra_defs = [a for a in sys.argv[1:] if a.startswith("-D")]
for n in [int(x) for x in sys.argv[1:] if not x.startswith("-D")]:
    objs = []
    for src, extra in B.SOURCES:
        op = os.path.join(B.OBJ, src.rsplit(".", 1)[0] + ".o")
        if src == "convp.hip":
            op = os.path.join(B.OBJ, f"pabl{n}_convp.o")
            subprocess.check_call([B._hipcc()] + B.COMMON + extra + extra_defs + [f"-DSMAP_CONVP_ABLATE={n}", "-c", os.path.join(B.CSRC, src), "-o", op])
        objs.append(op)
    out = os.path.join(B.OBJ, f"libsmap_hip_pabl{n}.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)
