#!/usr/bin/env python3
"""Build the REFERENCE's own `dapalib` extension for gfx950 as a checker (oracle/_ref/).

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the authoring container);
the GPU box just loads the prebuilt oracle/_ref/*.so that travel with the repo snapshot.

Recipe (what the reference's extensions/setup.py does through torch's CUDAExtension on a ROCm
PyTorch, spelled out; the reference's build system itself is not run):
  1. the image's own /opt/rocm/bin/hipify-perl rewrites the CUDA runtime spellings of the four
     sources that make up the path (association.cpp, arraygpu.hpp, gpu/nmsBase.cu,
     gpu/bodyPartConnectorBase.cu, gpu/cuda_cal.h) into a throw-away temp directory -- no
     hand-written stand-in for any header or library is involved (thrust = the image's rocThrust);
     gpu/cuda_cal.cu is dead code on this path and is not compiled;
  2. hipcc compiles them against the installed torch / pybind11 headers;
  3. ONLY the resulting shared objects land in oracle/_ref/ (git-ignored); the temp dir with
     the translated sources is deleted.  No reference source is copied into the repository.
Two variants:
  dapalib_ref.so        default hipcc flags (FMA contraction on, like nvcc's default)
  dapalib_ref_nofma.so  -ffp-contract=off : the float convention of oracle/smap_oracle.c and of
                        smap_amd/csrc/assoc.hip, so these three must agree bit for bit.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("SMAP_REFERENCE", "/root/reference")
FILES = ["association.cpp", "arraygpu.hpp", "gpu/nmsBase.cu", "gpu/bodyPartConnectorBase.cu", "gpu/cuda_cal.h"]
VARIANTS = {"dapalib_ref": [], "dapalib_ref_nofma": ["-ffp-contract=off"]}


def stale(target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(REF, "extensions", f)) > t for f in FILES) or \
        os.path.getmtime(__file__) > t


def main(force=False):
    ext = os.path.join(REF, "extensions")
    if not os.path.isdir(ext):
        print("build_ref: no reference checkout, nothing to do")
        return 0
    os.makedirs(OUT, exist_ok=True)
    todo = [n for n in VARIANTS if force or stale(os.path.join(OUT, n + ".so"))]
    if not todo:
        print("build_ref: up to date")
        return 0
    import torch
    from torch.utils import cpp_extension as ce
    tmp = tempfile.mkdtemp(prefix="smap_refbuild_")
    try:
        for f in FILES:
            dst = os.path.join(tmp, f)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            with open(dst, "w") as o:
                subprocess.check_call(["/opt/rocm/bin/hipify-perl", os.path.join(ext, f)], stdout=o,
                                      stderr=subprocess.DEVNULL)
        inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        for name in todo:
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip",
                   "-w", f"-DTORCH_EXTENSION_NAME={name}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                   "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                   f"-I{tmp}", f"-I{tmp}/gpu"] + [f"-I{i}" for i in inc] + VARIANTS[name] + \
                  [os.path.join(tmp, "association.cpp"), os.path.join(tmp, "gpu/nmsBase.cu"),
                   os.path.join(tmp, "gpu/bodyPartConnectorBase.cu"),
                   f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
                   f"-Wl,-rpath,{libdir}", "-o", os.path.join(OUT, name + ".so")]
            print("build_ref:", name, flush=True)
            subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(force="--force" in sys.argv))
