#!/usr/bin/env python3
"""Join a rocprofv3 kernel trace (rocpd sqlite .db) of bench.py with the engine's op list:
per-op duration, TFLOP/s and GB/s (algorithmic bytes: in + out + weights + residual/adds).

    python tools/prof_layers.py gpurun_out/prof/smap_results.db [B]
"""
import sqlite3
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    db = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch
    from types import SimpleNamespace as NS
    from smap_amd.engine import Graph, OP_CONV, OP_STEM, OP_MAXPOOL, OP_UPADD, OP_HEADSUM, OP_STEMPOOL, OP_TAPSUM, TILES
    from smap_amd.model.smap import SMAP
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
             OUTPUT_SHAPE=(128, 208), LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    g = Graph(SMAP(cfg).state_dict(), B, 512, 832, precision=os.environ.get("SMAP_PRECISION", "f16"))
    # kernel-name fragments per op kind (conv ops run conv.hip, conv2.hip, conv3.hip or conv1.hip kernels, by tile id)
    names = {OP_CONV: ("conv_igemm", "conv3x3_halo", "conv1x1_ws", "convp_kernel", "conv3_tail_kernel", "bottleneck_kernel", "bottleneck_first_kernel", "bottleneck128_kernel"), OP_STEM: ("stem_kernel",), OP_STEMPOOL: ("stem_pool_kernel",), OP_MAXPOOL: ("maxpool",),
             OP_UPADD: ("upadd",), OP_HEADSUM: ("headsum",), OP_TAPSUM: ("tapsum",)}       # set SMAP_NO_UPADD_FUSION=1 for pre-fusion traces
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    mine = [r for r in rows if any(k in r[0] for ks in names.values() for k in ks)]
    n = len(g.ops)
    assert len(mine) % n == 0, (len(mine), n)
    runs = len(mine) // n
    acc = [[] for _ in range(n)]
    for r in range(runs):
        for i in range(n):
            k = mine[r * n + i]
            assert any(f in k[0] for f in names[g.ops[i].kind]), (i, k[0])
            acc[i].append((k[2] - k[1]) / 1e3)
    tot = 0
    agg = {}
    buckets = {}
    print(f"{'op':4} {'name':44} {'shape':34} {'tile':8} {'us':>8} {'TF/s':>7} {'GB/s':>7}")
    for i, op in enumerate(g.ops):
        d = sorted(acc[i])[len(acc[i]) // 2]
        tot += d
        fl = by = 0
        shape = tile = ""
        if op.kind == OP_CONV:
            p, x, y = op.p, op.inp, op.out
            M = B * y.H * y.W
            K = p["ksize"] ** 2 * p["Cin"]
            fl, by = p["flops"], p["alg_bytes"]          # what the LAUNCH does: a whole-block op counts its three (four) convs
            shape = f"M{M} N{p['Cout']} K{K} k{p['ksize']}s{p['stride']}" + ("" if p["kinds"] in ("1x1", "3x3") else " " + p["kinds"])
            bk = ("whole-block launches" if p["kinds"] == "block" else f"{p['kinds']} s{p['stride']}") + f" at {y.H}x{y.W}"
            b_ = buckets.setdefault(bk, [0, 0.0, 0.0, 0.0])
            b_[0] += 1; b_[1] += d; b_[2] += fl; b_[3] += by
            tile = "x".join(map(str, TILES[p["tile"]])) + f"#{p['tile']}"
            key = (shape, tile)
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += d; a[2] += fl; a[3] += by
        name = op.out.name if op.out is not None else ("tapsum" if op.kind == OP_TAPSUM else "headsum")
        print(f"{i:4d} {name[-44:]:44} {shape:34} {tile:8} {d:8.1f} {fl / d / 1e6 if d else 0:7.1f} {by / d / 1e3 if d else 0:7.0f}")
    print("total us per forward", tot, "runs", runs)
    print("\n== by shape")
    for (shape, tile), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{shape:36} {tile:8} n={a[0]:3d} us={a[1]:8.1f} TF/s={a[2] / a[1] / 1e6:7.1f} GB/s={a[3] / a[1] / 1e3:7.0f}")


    conv_us = sum(b[1] for b in buckets.values())
    x3 = 3 if g.x3 else 1
    print(f"\n== buckets (conv launches: {sum(b[0] for b in buckets.values())}, {conv_us:.1f} us per forward; TF/s algorithmic, pipe = x{x3} / 2500)")
    for k, b in sorted(buckets.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:34} n={b[0]:3d} us={b[1]:8.1f} {100 * b[1] / conv_us:5.1f} %  TF/s={b[2] / b[1] / 1e6:7.1f} pipe={x3 * b[2] / b[1] / 1e6 / 2500:5.3f} GB/s={b[3] / b[1] / 1e3:7.0f}")


if __name__ == "__main__":
    main()
