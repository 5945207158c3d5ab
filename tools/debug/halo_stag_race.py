#!/usr/bin/env python3
"""Race screen for the staggered halo 3x3 schedule (tiles 44 / 45): full-size launches (every CU busy, other streams optional) must
reproduce the lockstep tiles 41 / 43 BIT FOR BIT (same accumulation order), run after run.

    python tools/debug/halo_stag_race.py [--runs 30] [--precision x3]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from smap_amd import lib as L  # noqa: E402
from bench_conv import build  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=30)
    ap.add_argument("--precision", choices=("f16", "x3"), default="x3")
    args = ap.parse_args()
    x3 = args.precision == "x3"
    dev = torch.device("cuda:0")
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream()
    noise = torch.randn(64 << 20, device=dev)
    bad = 0
    for shape in [(16, 32, 52, 256, 256), (16, 64, 104, 128, 128), (8, 16, 26, 512, 512), (3, 37, 45, 64, 192)]:
        for ref_t, t in ((41, 44), (43, 45)):
            outs = {}
            for tile in (ref_t, t):
                torch.manual_seed(1234)
                _, h, arena, blob, _, _ = build(*shape, 3, 1, tile, 0, dev, x3=x3)
                res = []
                for r in range(args.runs if tile == t else 1):
                    with torch.cuda.stream(side):                        # memory traffic from another stream while the conv runs
                        noise.mul_(1.0001)
                    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
                    torch.cuda.synchronize()
                    res.append(arena.clone())
                lib.smap_plan_destroy(h)
                outs[tile] = res
            diff = sum(int(not torch.equal(outs[ref_t][0], o)) for o in outs[t])
            bad += diff
            print(shape, f"tile {t} vs {ref_t}: {diff} of {len(outs[t])} runs differ", flush=True)
    print("RACE SCREEN", "FAILED" if bad else "clean")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
