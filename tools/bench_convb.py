#!/usr/bin/env python3
"""One whole-Bottleneck op (csrc/convb.hip) at the layer1 size, timed with HIP events over N back-to-back launches:
    [SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_convb<N>.so] python tools/bench_convb.py [tile ...] [--n 20]
With the ablation builds of tools/build_ablate.py --convb this decomposes the launch time (loads / MFMA / stores / weights)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smap_amd import lib as L  # noqa: E402
from smap_amd.engine import Graph  # noqa: E402
from smap_amd.model.smap import SMAP  # noqa: E402
from types import SimpleNamespace as NS  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 20
    tiles = [int(t) for t in args] or [91, 90, 93, 92]
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)), OUTPUT_SHAPE=(128, 208),
             LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    sd = SMAP(cfg).state_dict()
    lib = L.load()
    dev = torch.device("cuda:0")
    for tile in tiles:
        first = tile in (92, 93)
        os.environ["SMAP_BLOCK"], os.environ["SMAP_BLOCK_FIRST"] = ("64:91", f"64:{tile}") if first else (f"128:{tile}" if tile == 94 else f"64:{tile}", "")
        g = Graph(sd, 8, 512, 832, precision="x3")
        g.allocate()
        ops = g.emit()
        idx = next(i for i, op in enumerate(g.ops) if "head" in op.p and (("short" in op.p) == first))       # (tile 94: layer2's first identity block)
        one = (L.SmapOp * 1)(ops[idx])
        h = C.c_void_p()
        L.check(lib.smap_plan_create(one, 1, C.byref(h)), "create")
        arena = torch.randn(g.arena_bytes // 2, dtype=torch.float16, device=dev).mul_(0.1).view(torch.uint8)
        blob = g.weight_blob().to(dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        run = lambda: L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
        sums = set()
        for _ in range(3 if "--check" not in sys.argv else 12):     # --check: the launch must reproduce itself bit for bit (and other builds: compare the printed sums)
            run()
            if "--check" in sys.argv:
                sums.add(int(arena.view(torch.int64).sum().item()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        # (exported by diagnostics builds only: tools/build_ablate.py --one convb.hip SMAP_DEBUG_EXPORTS=1)
        occ = "n/a" if not hasattr(lib, "smap_debug_convb_occupancy") else (lib.smap_debug_convc_occupancy() if tile == 94 else lib.smap_debug_convb_occupancy(tile))
        if sums:
            print(f"   checksum(s) of the arena after each of 12 launches: {sorted(sums)}")
        print(f"{os.path.basename(L.SO_PATH):28s} occupancy {occ} tile {tile} ({g.ops[idx].out.name}): {e0.elapsed_time(e1) / n * 1e3:8.1f} us per launch (warm, back to back)")
        lib.smap_plan_destroy(h)


if __name__ == "__main__":
    main()
