#!/usr/bin/env python3
"""Build ablated copies of the library for bottleneck experiments (NOT shipped, NOT loaded by
default): smap_amd/csrc/obj/libsmap_hip_abl<N>.so with -DSMAP_ABLATE=N, selected by
SMAP_HIP_LIB=<path>.  N bits: 1 no K-loop loads, 2 no ds_read/MFMA, 4 no global stores, 8 no epilogue."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smap_amd import build as B  # noqa: E402

if "--convb" in sys.argv:       # csrc/convb.hip's identity kernel with parts switched off (SMAP_CONVB_ABLATE bits: 1 no x loads, 2 no MFMA,
    B.build_lib()                # 4 no stores, 8 no weight loads): libsmap_hip_convb<N>.so = the regular objects + an ablated convb.o
    for n in [x for x in sys.argv[sys.argv.index("--convb") + 1:]]:     # "3" = ablation bits; "lds64" = 64 KiB of LDS per workgroup
        op = os.path.join(B.OBJ, f"convb_abl{n}.o")
        flag = f"-DSMAP_CONVB_LDS_KB={n[3:]}" if n.startswith("lds") else f"-DSMAP_CONVB_ABLATE={n}"
        subprocess.check_call([B._hipcc()] + B.COMMON + [flag, "-c", os.path.join(B.CSRC, "convb.hip"), "-o", op])
        objs = [os.path.join(B.OBJ, src.rsplit(".", 1)[0] + ".o") if src != "convb.hip" else op for src, _ in B.SOURCES]
        out = os.path.join(B.OBJ, f"libsmap_hip_convb{n}.so")
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
        print(out)
    sys.exit(0)
if "--one" in sys.argv:         # --one SRC NAME=VALUE[,NAME=VALUE] ...: libsmap_hip_<src>_<tag>.so = the regular objects + SRC compiled with those -D switches
    B.build_lib()
    src = sys.argv[sys.argv.index("--one") + 1]
    for spec in sys.argv[sys.argv.index("--one") + 2:]:
        tag = spec.replace("=", "").replace("SMAP_", "").replace(",", "_").lower()
        op = os.path.join(B.OBJ, f"{src.rsplit('.', 1)[0]}_{tag}.o")
        subprocess.check_call([B._hipcc()] + B.COMMON + ["-D" + d for d in spec.split(",")] + ["-c", os.path.join(B.CSRC, src), "-o", op])
        objs = [os.path.join(B.OBJ, s_.rsplit(".", 1)[0] + ".o") if s_ != src else op for s_, _ in B.SOURCES]
        out = os.path.join(B.OBJ, f"libsmap_hip_{src.rsplit('.', 1)[0]}_{tag}.so")
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
        print(out)
    sys.exit(0)
if "--conv3" in sys.argv:       # csrc/conv3.hip alone with SMAP_ABLATE bits (1 no LDS-DMA, 2 no MFMA, 8 no epilogue, 16 no ds_read, 32 no barriers in the
    B.build_lib()                # staggered loop): libsmap_hip_conv3abl<N>.so = the regular objects + an ablated conv3.o
    for n in sys.argv[sys.argv.index("--conv3") + 1:]:
        op = os.path.join(B.OBJ, f"conv3_abl{n}.o")
        flag = ["-D" + d for d in n[2:].split(",")] if n.startswith("D:") else [f"-DSMAP_ABLATE={n}"]     # "D:NAME=VALUE[,NAME=VALUE]": other switches of conv3.hip
        n = n[2:].replace("=", "").replace("SMAP_", "").replace(",", "_").lower() if n.startswith("D:") else n
        op = os.path.join(B.OBJ, f"conv3_abl{n}.o")
        subprocess.check_call([B._hipcc()] + B.COMMON + flag + ["-c", os.path.join(B.CSRC, "conv3.hip"), "-o", op])
        objs = [os.path.join(B.OBJ, src.rsplit(".", 1)[0] + ".o") if src != "conv3.hip" else op for src, _ in B.SOURCES]
        out = os.path.join(B.OBJ, f"libsmap_hip_conv3abl{n}.so")
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
        print(out)
    sys.exit(0)
if "--timeline" in sys.argv:    # per-workgroup start/end stamps of every conv launch: libsmap_hip_timeline.so (tools/trace_pipeline.py)
    objs = []
    for src, extra in B.SOURCES:
        op = os.path.join(B.OBJ, "tl_" + src.rsplit(".", 1)[0] + ".o")
        subprocess.check_call([B._hipcc()] + B.COMMON + extra + ["-DSMAP_TIMELINE=1", "-c", os.path.join(B.CSRC, src), "-o", op])
        objs.append(op)
    out = os.path.join(B.OBJ, "libsmap_hip_timeline.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)
    sys.exit(0)
if "--trace" in sys.argv:       # phase-stamp build: libsmap_hip_trace.so (ConvArgs.dbg <- env SMAP_TRACE_PTR)
    objs = []
    for src, extra in B.SOURCES:
        op = os.path.join(B.OBJ, "trace_" + src.rsplit(".", 1)[0] + ".o")
        subprocess.check_call([B._hipcc()] + B.COMMON + extra + ["-DSMAP_TRACE=1", "-c", os.path.join(B.CSRC, src), "-o", op])
        objs.append(op)
    out = os.path.join(B.OBJ, "libsmap_hip_trace.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)
    sys.exit(0)
if "--epi-depth" in sys.argv:   # FULL epilogue with 2 passes of loads in flight: libsmap_hip_epi2.so
    objs = []
    for src, extra in B.SOURCES:
        op = os.path.join(B.OBJ, src.rsplit(".", 1)[0] + ".o")
        if src == "conv.hip":
            op = os.path.join(B.OBJ, "epi2_conv.o")
            subprocess.check_call([B._hipcc()] + B.COMMON + extra + ["-DSMAP_EPI_DEPTH=2", "-c", os.path.join(B.CSRC, src), "-o", op])
        objs.append(op)
    out = os.path.join(B.OBJ, "libsmap_hip_epi2.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)
    sys.exit(0)
for n in [int(x) for x in sys.argv[1:]] or [1, 2, 8, 9, 10]:
    objs = []
    for src, extra in B.SOURCES:
        op = os.path.join(B.OBJ, f"abl{n}_" + src.rsplit(".", 1)[0] + ".o")
        subprocess.check_call([B._hipcc()] + B.COMMON + extra + [f"-DSMAP_ABLATE={n}", "-c", os.path.join(B.CSRC, src), "-o", op])
        objs.append(op)
    out = os.path.join(B.OBJ, f"libsmap_hip_abl{n}.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)
