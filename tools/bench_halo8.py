#!/usr/bin/env python3
"""Eight-wave halo 3x3 tiles (csrc/conv3.hip, ids 40..43) against the tiles the shipped table holds, per plain 3x3 shape of the
schedule, COLD (three rotating arenas, as inside the full schedule), through 1-op plans of the C ABI.

    python tools/bench_halo8.py [--batch 16] [--precision x3] [--iters 20]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from smap_amd import lib as L  # noqa: E402
from smap_amd.engine import TILES, _table_entry  # noqa: E402
from bench_conv import build  # noqa: E402

SHAPES = [(64, 104, 128, 128), (32, 52, 256, 256), (16, 26, 512, 512), (128, 208, 256, 43), (64, 104, 256, 43), (32, 52, 256, 43),
          (128, 208, 64, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--precision", choices=("f16", "x3"), default="x3")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="H,W,Cin,Cout: this shape only")
    ap.add_argument("--tiles", default="", help="comma-separated tile ids instead of the default candidate list (ablation builds: SMAP_HIP_LIB=...)")
    args = ap.parse_args()
    x3 = args.precision == "x3"
    table = json.load(open(os.path.join(ROOT, "smap_amd", "tile_table_x3.json" if x3 else "tile_table.json")))
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = L.load()
    shapes = [tuple(int(v) for v in args.only.split(","))] if args.only else SHAPES
    for H, W, Cin, Cout in shapes:
        key = f"{args.batch},{H},{W},{Cin},{Cout},3,1"
        shipped = _table_entry(table[key])[:2] if key in table else []
        new = [t for t in (40, 41, 42, 43, 44, 45) if not (Cout <= 64 and TILES[t][1] > 64) and not (Cout > 64 and TILES[t][1] == 64)]
        res = {}
        cands = shipped + [t for t in (31, 35, 34, 36, 52, 60) if t not in shipped and not (Cout <= 64 and TILES[t][1] > 64)
                           and not (Cout > 64 and TILES[t][1] == 64)] + new
        if args.tiles:
            cands = [int(t) for t in args.tiles.split(",")]
        for t in cands:
            try:
                _, h, arena, blob, flops, byts = build(args.batch, H, W, Cin, Cout, 3, 1, t, 0, dev, x3=x3)
            except L.SmapError:
                continue
            arenas = [arena, arena.clone(), arena.clone()]
            k = [0]

            def run():
                ar = arenas[k[0] % 3]
                k[0] += 1
                L.check(lib.smap_plan_run(h, None, C.c_void_p(ar.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[t] = round(e0.elapsed_time(e1) * 1e3 / args.iters, 1)
            lib.smap_plan_destroy(h)
            del arena, blob, arenas
        best = min(res, key=res.get)
        print(key, "shipped", shipped, res, "best", best, f"{flops * (3 if x3 else 1) / res[best] * 1e-6:.0f} TF/s on the pipe", flush=True)


if __name__ == "__main__":
    main()
