#!/usr/bin/env python3
"""Tile choice for the MERGED 1x1 launches of the schedule (smap_amd/engine.py::Graph.conv_seg: several convs on one input, one
launch, one output tensor each): every distinct merged op of the B-frame schedule is rebuilt alone with each conv.hip tile, run
cold (three rotating arenas) through the C ABI, and the ranked winners are added to smap_amd/tile_table_x3.json under the merged
key "B,H,W,Cin,c0+c1[+c2],1,1[,up]".

    python tools/autotune_seg.py --batch 8 --batch 16 [--iters 20] [--write]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smap_amd import lib as L  # noqa: E402
from smap_amd import engine as E  # noqa: E402


def single_op_graph(sd, B, x_shape, segs, up_shape, tile):
    """A Graph holding ONE merged launch: input tensor, optional low-resolution `up` tensor, the op."""
    g = E.Graph(sd, B, 512, 832, precision="x3", build=False)
    x = g.tensor("x", *x_shape)
    up = g.tensor("up", *up_shape) if up_shape else None
    g.conv_seg(segs, x, up=up, tile=tile)
    x.first = 0
    if up is not None:
        up.first = 0
    g.allocate(reuse=False)
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, action="append")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--write", action="store_true", help="merge the winners into smap_amd/tile_table_x3.json")
    ap.add_argument("--out", default="", help="write the merged table here instead (a GPU visit: gpurun_out/..., then SMAP_TILE_TABLE_X3=<path>)")
    args = ap.parse_args()
    from types import SimpleNamespace as NS
    from smap_amd.model.smap import SMAP
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
             OUTPUT_SHAPE=(128, 208), LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    sd = {k: v.detach().cpu() for k, v in SMAP(cfg).state_dict().items()}
    dev = torch.device("cuda:0")
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    path = os.path.join(ROOT, "smap_amd", "tile_table_x3.json")
    table = json.load(open(path))
    for B in (args.batch or [8, 16]):
        full = E.Graph(sd, B, 512, 832, precision="x3")
        seen = {}
        for op in full.ops:
            if not op.outs:
                continue
            couts = [op.p["Cout"]] + [sg["cout"] for sg in op.p["segs"]]
            key = f"{B},{op.inp.H},{op.inp.W},{op.p['Cin']},{'+'.join(map(str, couts))},1,1" + (",up" if op.aux else "")
            if key in seen:
                seen[key][0] += 1
                continue
            names = [op.out.name] + [t.name for t in op.outs]
            pres = {"out": "u_skip", "res1": "res_conv1", "up_conv@low": "up_conv"}
            segs = []
            for nm, relu in zip(names, [op.p["relu"]] + [sg["relu"] for sg in op.p["segs"]]):
                unit, leaf = nm.rsplit(".", 1)
                segs.append((nm, unit + "." + pres.get(leaf, leaf), bool(relu)))
            up_shape = (op.aux[0].H, op.aux[0].W, op.aux[0].C) if op.aux else None
            seen[key] = [1, (op.inp.H, op.inp.W, op.inp.C), segs, up_shape, op.p["tile"]]
        for key, (count, x_shape, segs, up_shape, cur) in seen.items():
            res = {}
            for t in [t for t in E.X3_TILES if E.tile_family(t) == "igemm" and t != 3]:
                try:
                    g = single_op_graph(sd, B, x_shape, segs, up_shape, t)
                    ops = g.emit()
                    h = C.c_void_p()
                    L.check(lib.smap_plan_create(ops, len(g.ops), C.byref(h)), "create")
                except (L.SmapError, AssertionError, StopIteration):
                    continue
                blob = g.weight_blob().to(dev)
                arenas = []
                for _ in range(3):
                    a = torch.zeros((g.arena_bytes,), dtype=torch.uint8, device=dev)
                    a[E.ZERO_PAGE:].view(torch.float16).copy_((torch.randn((g.arena_bytes - E.ZERO_PAGE) // 2, device=dev) * 0.5).to(torch.float16))
                    arenas.append(a)
                i = [0]

                def run():
                    ar = arenas[i[0] % 3]
                    i[0] += 1
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
                res[t] = e0.elapsed_time(e1) * 1e3 / args.iters
                lib.smap_plan_destroy(h)
                del arenas, blob
            ranked = sorted(res, key=res.get)
            print(key, f"x{count}", "shipped", cur, {t: round(res[t], 1) for t in ranked[:6]}, flush=True)
            table[key] = ranked[0]
    if args.write or args.out:
        json.dump(table, open(args.out or path, "w"), indent=0, sort_keys=True)
        print("wrote", args.out or path)


if __name__ == "__main__":
    main()
