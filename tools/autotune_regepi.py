#!/usr/bin/env python3
"""Round 6: where does the 256 x 256 register-epilogue tile (csrc/conv.hip, tile id 56) beat the tile the table holds?  Every distinct
conv launch of the B-frame split-precision schedule that the tile can take (N >= 256 after padding, fp16 output, no fused bilinear add, no
post-ReLU addends) is rebuilt alone -- plain launches through tools/bench_conv.py::build, merged 1x1 launches through
tools/autotune_seg.py::single_op_graph -- and timed cold (three rotating arenas) with its current tile and with tile 56.  Winners by at least
--gain go to the table as ranked lists [56, current ...] (ops of the same shape that the tile cannot take fall through).

    python tools/autotune_regepi.py --batch 16 --batch 8 [--iters 20] [--gain 0.03] --out gpurun_out/.../tile_table_x3.json
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
from smap_amd import engine as E  # noqa: E402
from bench_conv import build  # noqa: E402
from autotune_seg import single_op_graph  # noqa: E402

NEW = 56


def time_plan(lib, h, arenas, blob, st, iters):
    i = [0]

    def run():
        ar = arenas[i[0] % len(arenas)]
        i[0] += 1
        L.check(lib.smap_plan_run(h, None, C.c_void_p(ar.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, action="append")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gain", type=float, default=0.03)
    ap.add_argument("--out", default="")
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
    saved = 0.0
    for B in (args.batch or [16, 8]):
        full = E.Graph(sd, B, 512, 832, precision="x3")
        seen = {}
        for op in full.ops:
            if op.kind != E.OP_CONV or "head" in op.p or "tail" in op.p:
                continue
            p, x = op.p, op.inp
            if p["out_fp32"] or op.aux or op.add1 is not None or op.add2 is not None or p["cout_pad"] < 256 and not op.outs:
                continue
            if op.outs:
                couts = [p["Cout"]] + [sg["cout"] for sg in p["segs"]]
                if sum(couts) < 256:
                    continue
                key = f"{B},{x.H},{x.W},{p['Cin']},{'+'.join(map(str, couts))},1,1"
                names = [op.out.name] + [t.name for t in op.outs]
                pres = {"out": "u_skip", "res1": "res_conv1", "up_conv@low": "up_conv"}
                segs = []
                for nm, relu in zip(names, [p["relu"]] + [sg["relu"] for sg in p["segs"]]):
                    unit, leaf = nm.rsplit(".", 1)
                    segs.append((nm, unit + "." + pres.get(leaf, leaf), bool(relu)))
                info = ("seg", (x.H, x.W, x.C), segs)
            else:
                if p["Cout"] % 8 or p["Cout"] < 200:
                    continue
                key = f"{p['frames']},{x.H},{x.W},{p['Cin']},{p['Cout']},{p['ksize']},{p['stride']}"
                info = ("one", (p["frames"], x.H, x.W, p["Cin"], p["Cout"], p["ksize"], p["stride"]), int(op.res is not None))
            if key in seen:
                seen[key][0] += 1
            else:
                seen[key] = [1, p["tile"], info]
        for key, (count, cur, info) in seen.items():
            res = {}
            for t in (cur, NEW):
                try:
                    if info[0] == "seg":
                        g = single_op_graph(sd, B, info[1], info[2], None, t)
                        ops = g.emit()
                        h = C.c_void_p()
                        L.check(lib.smap_plan_create(ops, len(g.ops), C.byref(h)), "create")
                        blob = g.weight_blob().to(dev)
                        arenas = []
                        for _ in range(3):
                            a = torch.zeros((g.arena_bytes,), dtype=torch.uint8, device=dev)
                            a[E.ZERO_PAGE:].view(torch.float16).copy_((torch.randn((g.arena_bytes - E.ZERO_PAGE) // 2, device=dev) * 0.5).to(torch.float16))
                            arenas.append(a)
                    else:
                        lib, h, arena, blob, _, _ = build(*info[1], t, info[2], dev, x3=True)
                        arenas = [arena, arena.clone(), arena.clone()]
                except (L.SmapError, AssertionError, StopIteration) as e:
                    print(key, "tile", t, "rejected:", e)
                    continue
                res[t] = time_plan(lib, h, arenas, blob, st, args.iters)
                lib.smap_plan_destroy(h)
                del arenas, blob
            if cur in res and NEW in res:
                gain = 1.0 - res[NEW] / res[cur]
                take = gain >= args.gain
                print(f"{key:44s} x{count:2d}  tile {cur:2d}: {res[cur]:7.1f} us   tile {NEW}: {res[NEW]:7.1f} us   {100 * gain:+5.1f} %  {'TAKE' if take else ''}", flush=True)
                if take:
                    old = table.get(key, cur)
                    table[key] = [NEW] + [t for t in (old if isinstance(old, list) else [old]) if t != NEW]
                    saved += (res[cur] - res[NEW]) * count * (8.0 / B)
    print(f"isolated saving per 8 frames (both batch sizes summed): {saved:.0f} us")
    if args.out:
        json.dump(table, open(args.out, "w"), indent=0, sort_keys=True)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
