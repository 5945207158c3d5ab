#!/usr/bin/env python3
"""The memory-path model of DESIGN.md section 6, evaluated per op and set against a measured per-layer file (tools/prof_layers.py
output of a --depth 1 --launch-frames 0 trace).  No GPU needed: the schedule is rebuilt on the CPU, the measurements are read.

    time(op) ~= max( fabric_read / R + written / W + l2_to_lds / S ,  mfma / P ) + launch
      fabric_read : unique input bytes + residual / skip / bilinear-tap tensors + weights x min(8 XCDs, workgroups)
      written     : output bytes
      l2_to_lds   : bytes the workgroups stage into LDS: per output tile (BM + BN) x K x 2 x planes (im2col stages a tap's
                    channels per K tile; the halo kernel stages its patch once per channel chunk and a weight tile per tap)
      mfma        : 3 x (or 1 x) algorithmic FLOPs at the dense fp16 peak, derated by the share of CUs the launch can occupy

    python tools/model_memory_path.py profiles/r3_final_x3_layers.txt [B] > profiles/r3_model_vs_measured.txt
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, W, S, P, LAUNCH, CUS = 5.3e12, 6.2e12, 21e12, 2.5e15, 4e-6, 256      # B/s, B/s, B/s, FLOP/s, s


def main():
    path = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch
    from types import SimpleNamespace as NS
    from smap_amd.engine import Graph, OP_CONV, TILES, tile_family
    from smap_amd.model.smap import SMAP
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
             OUTPUT_SHAPE=(128, 208), LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    g = Graph(SMAP(cfg).state_dict(), B, 512, 832, precision="x3")
    meas = {}
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+(\S+)\s+(M\d+ N\d+ K\d+ k\ds\d(?: [a-z0-9+]+)?)\s+(\S+)#(\d+)\s+([\d.]+)", line)
        if m:
            meas[int(m.group(1))] = (m.group(2), int(m.group(5)), float(m.group(6)))
    rows, tot_m, tot_p, parts = [], 0.0, 0.0, [0.0, 0.0, 0.0, 0.0]
    for i, op in enumerate(g.ops):
        if op.kind != OP_CONV or i not in meas:
            continue
        p, x, y = op.p, op.inp, op.out
        name, tile, us = meas[i]
        if tile != p["tile"] or not y.name.endswith(name[-30:]):
            raise SystemExit(f"op {i}: the trace ran tile {tile} ({name}), the schedule here picks {p['tile']} ({y.name}) -- same tile table?")
        nfr, k, cin, cout = p["frames"], p["ksize"], p["Cin"], p["Cout"]
        M = nfr * y.H * y.W
        K = k * k * cin
        planes = 2
        bm, bn = TILES[tile]
        mt, nt = -(-M // bm), p["cout_pad"] // bn
        wbytes = p["cout_pad"] * K * 2 * planes
        inb = nfr * (y.H * y.W if k == 1 else x.H * x.W) * cin * 2 * planes        # a strided 1x1 touches the sampled pixels only
        extra = sum(t.nbytes * nfr // g.B for t in (op.res, op.add1, op.add2) if t is not None)
        if op.aux:
            extra += 4 * y.nbytes * nfr // g.B // (1 if y.esize == 2 else 1)          # four taps per output element (mostly L2 hits: counted once more below)
            extra = extra - 3 * y.nbytes * nfr // g.B                                    # fabric: the low-res tensor is 1/4 of the output
        fabric = inb + extra + wbytes * min(8, mt * nt)
        written = nfr * y.H * y.W * y.C * y.esize * (planes if y.esize == 2 else 1)
        flops = p.get("flops", 2.0 * M * cout * K)
        if p.get("kinds") == "block":              # whole Bottleneck (convb / convc): x patch with halo once, every weight of the block per workgroup
            hc = p["head"]["cin"]
            th, tw = (4, 16) if tile in (90, 92) else (8, 16)
            wgs = -(-y.W // tw) * -(-y.H // th) * nfr
            wall = (cin * hc + 9 * cin * cin + y.C * cin + (y.C * hc if "short" in p else 0)) * 2 * planes
            inb = nfr * x.H * x.W * hc * 2 * planes
            fabric = inb + extra - (op.res.nbytes if op.res is not None else 0) + wall * min(8, wgs)    # the residual IS the input: read once
            l2lds = wgs * ((th + 2) * (tw + 2) * hc * 2 * planes + wall)
        elif tile_family(tile) == "halo":
            tw = 16 if tile in (30, 31, 34, 35, 38) else 32
            th = 128 // tw
            prow = -(-((th + 2) * (tw + 2)) // 32) * 32
            wgs = -(-y.W // tw) * -(-y.H // th) * nfr * nt
            l2lds = wgs * (cin // 32) * (prow * 128 + 9 * bn * 128)
        else:
            wgs = mt * nt
            l2lds = wgs * (bm + bn) * K * 2 * planes
        if op.aux:
            l2lds += 4 * written                                                         # the bilinear taps come through the same path
        t_mem = fabric / R + written / W + l2lds / S
        t_mfma = 3 * flops / (P * min(1.0, wgs / CUS))
        t = max(t_mem, t_mfma) + LAUNCH
        rows.append((i, y.name[-42:], f"M{M} N{cout} K{K}", tile, us, t * 1e6, fabric / R * 1e6, written / W * 1e6, l2lds / S * 1e6, t_mfma * 1e6))
        tot_m += us
        tot_p += t * 1e6
        for j, v in enumerate((fabric / R, written / W, l2lds / S, t_mfma)):
            parts[j] += v * 1e6
    print(f"{'op':>4} {'name':42} {'shape':26} {'tile':>4} {'meas us':>8} {'model':>8} {'ratio':>6} | {'reads':>7} {'writes':>7} {'L2->LDS':>8} {'mfma':>7}")
    for r in rows:
        print(f"{r[0]:4d} {r[1]:42} {r[2]:26} {r[3]:4d} {r[4]:8.1f} {r[5]:8.1f} {r[4] / r[5]:6.2f} | {r[6]:7.1f} {r[7]:7.1f} {r[8]:8.1f} {r[9]:7.1f}")
    import statistics
    ratios = [r[4] / r[5] for r in rows]
    print(f"\n{len(rows)} conv launches: measured {tot_m:.0f} us, model {tot_p:.0f} us (ratio {tot_m / tot_p:.2f}); per-op ratio median "
          f"{statistics.median(ratios):.2f}, 10 % / 90 % quantiles {sorted(ratios)[len(ratios) // 10]:.2f} / {sorted(ratios)[9 * len(ratios) // 10]:.2f}")
    print(f"model terms summed over the launches (us): fabric reads {parts[0]:.0f}, writes {parts[1]:.0f}, L2->LDS {parts[2]:.0f}; MFMA at peak {parts[3]:.0f}")
    print(f"constants: reads {R / 1e12} TB/s, writes {W / 1e12} TB/s, L2->LDS {S / 1e12} TB/s, MFMA {P / 1e15} PFLOP/s x min(1, workgroups / {CUS}), launch {LAUNCH * 1e6} us")


if __name__ == "__main__":
    main()
