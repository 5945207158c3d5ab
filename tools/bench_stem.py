#!/usr/bin/env python3
"""The stem launch (7x7 s2 conv + BN + ReLU, csrc/plan.hip::stem_kernel) of the 8-frame split-precision schedule, timed alone over N
back-to-back launches through a 1-op plan:  [SMAP_HIP_LIB=...] python tools/bench_stem.py [--n 30]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smap_amd import lib as L  # noqa: E402
from smap_amd.engine import Graph, OP_STEM  # noqa: E402
from smap_amd.model.smap import SMAP  # noqa: E402
from types import SimpleNamespace as NS  # noqa: E402


def main():
    n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 30
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)), OUTPUT_SHAPE=(128, 208),
             LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    g = Graph(SMAP(cfg).state_dict(), 8, 512, 832, precision="x3")
    g.allocate()
    ops = g.emit()
    idx = next(i for i, op in enumerate(g.ops) if op.kind == OP_STEM)
    one = (L.SmapOp * 1)(ops[idx])
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.smap_plan_create(one, 1, C.byref(h)), "create")
    dev = torch.device("cuda:0")
    arena = torch.zeros(g.arena_bytes, dtype=torch.uint8, device=dev)
    blob = g.weight_blob().to(dev)
    img = torch.rand(8, 3, 512, 832, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: L.check(lib.smap_plan_run(h, C.c_void_p(img.data_ptr()), C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    out = g.ops[idx].out
    print(f"{os.path.basename(L.SO_PATH):40s} stem 8 x 3x512x832 -> {out.H}x{out.W}x{out.C}: {e0.elapsed_time(e1) / n * 1e3:7.1f} us per launch; "
          f"checksum {int(arena.view(torch.int64).sum().item())}")
    lib.smap_plan_destroy(h)


if __name__ == "__main__":
    main()
