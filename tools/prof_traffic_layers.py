#!/usr/bin/env python3
"""Per-op HBM traffic of one B = 8 forward from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only,
`bench.py --depth 1 --steps 2 --warmup 1`): counter bytes next to the algorithmic bytes of every conv op, so that the
over-fetch (activation tiles re-read across N tiles, weight tiles missing L2) can be pinned on layers.
    python tools/prof_traffic_layers.py fetch.csv write.csv [precision]
FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, calibrated in round 1)."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERN = ("conv_igemm", "conv3x3_halo", "convp_kernel", "stem_kernel", "stem_pool_kernel", "maxpool", "upadd", "headsum")


def last_forward(path, counter, n_ops):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERN)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rows) % n_ops == 0, (len(rows), n_ops)
    return rows[-n_ops:]


def main():
    prec = sys.argv[3] if len(sys.argv) > 3 else "x3"
    import torch
    from types import SimpleNamespace as NS
    from smap_amd.engine import Graph, OP_CONV, TILES
    from smap_amd.model.smap import SMAP
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
             OUTPUT_SHAPE=(128, 208), LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    g = Graph(SMAP(cfg).state_dict(), 8, 512, 832, precision=prec)
    n = len(g.ops)
    fr, wr = last_forward(sys.argv[1], "FETCH_SIZE", n), last_forward(sys.argv[2], "WRITE_SIZE", n)
    tot = [0.0] * 4
    agg = {}
    print(f"{'op':4} {'name':40} {'shape':30} {'tile':5} {'rd MB':>8} {'alg':>8} {'x':>5} {'wr MB':>8} {'alg':>8}")
    for i, op in enumerate(g.ops):
        rd, wrb = 2.0 * float(fr[i]["Counter_Value"]) * 1024, float(wr[i]["Counter_Value"]) * 1024
        if op.kind != OP_CONV:
            continue
        p, x, y = op.p, op.inp, op.out
        K = p["ksize"] ** 2 * p["Cin"]
        frames = p["frames"]
        a_rd = frames * x.H * x.W * p["Cin"] * 2 * x.planes + p["cout_pad"] * K * 2 * (2 if g.x3 else 1)
        a_rd += sum(t.nbytes for t in (op.res, op.add1, op.add2) if t is not None)
        a_rd += op.aux[0].nbytes if op.aux else 0
        a_wr = y.nbytes * frames // g.B
        shape = f"M{frames * y.H * y.W} N{p['Cout']} K{K} k{p['ksize']}s{p['stride']}"
        print(f"{i:4d} {y.name[-40:]:40} {shape:30} {p['tile']:5d} {rd / 1e6:8.1f} {a_rd / 1e6:8.1f} {rd / a_rd:5.2f} {wrb / 1e6:8.1f} {a_wr / 1e6:8.1f}")
        for k, v in enumerate((rd, a_rd, wrb, a_wr)):
            tot[k] += v
        a = agg.setdefault((shape, p["tile"]), [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += rd; a[2] += a_rd; a[3] += wrb; a[4] += a_wr
    print(f"conv total: read {tot[0] / 1e9:.2f} GB (algorithmic {tot[1] / 1e9:.2f}), written {tot[2] / 1e9:.2f} GB (algorithmic {tot[3] / 1e9:.2f})")
    print("\n== by shape, sorted by excess read bytes")
    for (shape, tile), a in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2])):
        print(f"{shape:32} #{tile:<3d} n={a[0]:3d} read {a[1] / 1e6:8.1f} MB vs {a[2] / 1e6:8.1f} (x{a[1] / a[2]:.2f}, excess {(a[1] - a[2]) / 1e6:7.1f} MB)  written {a[3] / 1e6:8.1f} vs {a[4] / 1e6:8.1f}")


if __name__ == "__main__":
    main()
