"""Build libsmap_hip.so (the gfx950 HIP library behind include/smap_hip.h) in-tree.

    python -m smap_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels
with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsmap_hip.so")
OBJ = os.path.join(CSRC, "obj")

# (source, extra flags).  assoc.hip is bit-exact float work: contraction OFF.
SOURCES = [
    ("assoc.hip", ["-ffp-contract=off"]),
    ("conv.hip", []),
    ("conv3.hip", []),
    ("convp.hip", []),
    ("convf.hip", []),
    ("convb.hip", []),
    ("convc.hip", []),
    ("plan.hip", []),
]
# -fvisibility=hidden: the .so exports exactly what include/smap_hip.h declares (its visibility push / pop), nothing of the internals
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
          "-I" + CSRC, "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build_lib(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(ROOT, "include", "smap_hip.h")] + [
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".cuh"))]
    objs, relink = [], force
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        if force or _newer(sp, op) or any(_newer(h, op) for h in headers):
            cmd = [hipcc] + COMMON + extra + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
        objs.append(op)
    if relink or not os.path.exists(OUT):
        # the dynamic symbol table = the functions include/smap_hip.h declares (-fvisibility=hidden + the header's visibility push) and
        # nothing else: a linker version script also keeps hipcc's per-unit __hip_cuid_* markers and weak template instantiations local
        vmap = os.path.join(OBJ, "exports.map")
        with open(vmap, "w") as f:
            f.write("{ global: smap_*; local: *; };\n")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vmap, "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
