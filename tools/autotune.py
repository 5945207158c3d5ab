#!/usr/bin/env python3
"""Pick the fastest conv tile variant per distinct layer shape of the SMAP schedule by measuring
every candidate on the GPU (single-op plans through the C ABI), and write the table that
smap_amd/engine.py::pick_tile consults:  smap_amd/tile_table.json  {"B,H,W,Cin,Cout,k,s": tile}.

    python tools/autotune.py [--batch 8] [--iters 20]
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
from smap_amd.engine import Graph, OP_CONV, TILES  # noqa: E402
from bench_conv import build  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "smap_amd", "tile_table.json"))
    ap.add_argument("--precision", choices=("f16", "x3"), default="f16",
                    help="x3: tune the split-precision instances (-> smap_amd/tile_table_x3.json unless --out is given)")
    ap.add_argument("--halo", type=float, default=0.0, metavar="GAIN",
                    help="only revisit the shapes the specialised kernels cover: keep the tile of the existing table unless "
                         "a halo-tiled 3x3 (csrc/conv3.hip, ids 30..39) or weight-stationary 1x1 (csrc/conv1.hip, ids "
                         "40..41) variant is at least GAIN (e.g. 0.05) faster")
    ap.add_argument("--convp", type=float, default=-1.0, metavar="GAIN",
                    help="only measure the persistent wave-specialised tiles (csrc/convp.hip, ids 60..62) against the tile "
                         "the existing table holds, cold (three rotating arenas); a shape's entry becomes the ranked list "
                         "[convp tile, old tile] when the convp tile is at least GAIN faster (ops it cannot take -- fused "
                         "bilinear add, fp32 output -- fall through to the old tile)")
    ap.add_argument("--cold", action="store_true", help="three rotating arenas per candidate: every launch finds its operands cold in "
                    "the Infinity Cache, as inside the full schedule (the warm loop flatters HBM-shaped layers by 5-20 %%)")
    ap.add_argument("--table", default="", help="existing table to start from (--convp / --halo); default: the shipped one")
    args = ap.parse_args()
    x3 = args.precision == "x3"
    if x3 and args.out == os.path.join(ROOT, "smap_amd", "tile_table.json"):
        args.out = os.path.join(ROOT, "smap_amd", "tile_table_x3.json")
    from types import SimpleNamespace as NS
    from smap_amd.model.smap import SMAP
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
             OUTPUT_SHAPE=(128, 208), LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    g = Graph(SMAP(cfg).state_dict(), args.batch, 512, 832)
    shapes = {}
    for op in g.ops:
        if op.kind != OP_CONV:
            continue
        p, x = op.p, op.inp
        key = (args.batch, x.H, x.W, p["Cin"], p["Cout"], p["ksize"], p["stride"])
        fused = op.res is not None or op.add1 is not None or op.add2 is not None or bool(op.aux)
        shapes.setdefault(key, [0, op.res is not None, True, 0])
        shapes[key][0] += 1
        shapes[key][2] &= not fused                      # halo kernel: plain epilogue only
        shapes[key][3] += int(not op.aux and not p["out_fp32"] and p["Cout"] % 8 == 0)    # ops convp.hip can take
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    table, total_best, total_default = {}, 0.0, 0.0
    convp = args.convp >= 0
    shipped = args.table or os.path.join(ROOT, "smap_amd", "tile_table_x3.json" if x3 else "tile_table.json")
    old = json.load(open(shipped)) if (args.halo or convp) else {}
    table.update(old)
    for key, (count, has_res, plain, n_convp) in sorted(shapes.items()):
        B, H, W, Cin, Cout, k, s = key
        halo_ok = plain and k == 3 and s == 1
        skey = ",".join(map(str, key))
        ws_ok = False      # the weight-stationary 1x1 kernel left the product build (tools/experiments/)
        if args.halo and not ((halo_ok or ws_ok) and skey in old):
            continue
        if Cout <= 32:
            cands = [3, 8, 38, 39]
        elif Cout <= 64:
            cands = [t for t, (bm, bn) in TILES.items() if bn == 64]
        else:
            cands = [t for t, (bm, bn) in TILES.items() if bn >= 64]
        cands = [t for t in cands if t < 30 or (halo_ok and t < 40)]
        if x3:
            from smap_amd.engine import X3_TILES
            cands = [3] if Cout <= 32 else [t for t in X3_TILES if not (Cout <= 64 and TILES[t][1] > 64) and t not in (66, 68)
                                            and not (Cout > 64 and TILES[t][1] == 64 and t >= 60)]
            if halo_ok:      # halo-tiled 3x3 kernel has split-precision instances too
                cands += [38, 39] if Cout <= 32 else [t for t in range(30, 38) if not (Cout <= 64 and TILES[t][1] > 64)]
        if args.halo:
            cands = [old[skey]] + [t for t in cands if t >= 30] + ([40, 41] if ws_ok else [])
        if convp:
            if not n_convp or skey not in old or Cout <= 32:
                continue
            old_t = old[skey][-1] if isinstance(old[skey], list) else old[skey]
            cands = [old_t] + [t for t in (60, 61, 62) if not (Cout <= 64 and TILES[t][1] > 64) and
                               (Cout + TILES[t][1] - 1) // TILES[t][1] * TILES[t][1] <= 2048]
        res = {}
        for t in cands:
            try:
                lib, h, arena, blob, flops, byts = build(B, H, W, Cin, Cout, k, s, t, int(has_res), dev, x3=x3)
            except L.SmapError:                         # the plan rejects this tile for this op
                continue
            arenas = [arena] + ([arena.clone(), arena.clone()] if (convp or args.cold) else [])
            def run(i=[0]):
                ar = arenas[i[0] % len(arenas)]
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
            del arena, blob, arenas
        best = min(res, key=res.get)
        if convp:
            old_t = cands[0]
            print(key, "x%d (%d convp-able)" % (count, n_convp), {t: round(v, 1) for t, v in res.items()}, "best", best, flush=True)
            if best != old_t and res[best] <= (1.0 - args.convp) * res[old_t]:
                table[skey] = [best, old_t]
                total_best += res[best] * n_convp + res[old_t] * (count - n_convp)
            else:
                total_best += res[old_t] * count
            total_default += res[old_t] * count
            continue
        if args.halo and res[best] > (1.0 - args.halo) * res[old[skey]]:
            best = old[skey]
        from smap_amd.engine import pick_tile_heuristic, pick_tile_x3
        Mo = B * ((H + 2 * (k // 2) - k) // s + 1) * ((W + 2 * (k // 2) - k) // s + 1)
        dflt = pick_tile_x3(Mo, Cout)[-1] if x3 else pick_tile_heuristic(Mo, Cout)
        ranked = sorted(res, key=res.get)
        from smap_amd.engine import tile_family
        entry = []
        for t in ranked:                                 # ranked list down to the first tile that takes every op of the shape
            entry.append(t)
            if tile_family(t) == "igemm":
                break
        table[",".join(map(str, key))] = entry[0] if len(entry) == 1 else entry
        total_best += res[best] * count
        total_default += res.get(dflt, res[best]) * count
        print(key, "x%d" % count, {t: round(v, 1) for t, v in res.items()}, "best", best, "default", dflt, flush=True)
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print(f"sum best {total_best:.0f} us vs heuristic {total_default:.0f} us -> {args.out}")


if __name__ == "__main__":
    main()
