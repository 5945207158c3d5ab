#!/usr/bin/env python3
"""Single-layer micro-benchmark of the MFMA conv kernel (through the C ABI, 1-op plans).

    python tools/bench_conv.py [--iters 50] [--only L2,L4] [--tile-override L2:0]
Prints us / TFLOP/s / algorithmic GB/s per preset (B=8 shapes of the SMAP backbone).
Wrap in `rocprofv3 --pmc ...` for counters (use --iters 3)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from smap_amd import lib as L  # noqa: E402
from smap_amd.engine import TILES  # noqa: E402

PRESETS = {          # B, H, W, Cin, Cout, k, stride, tile, residual
    "L1": (8, 128, 208, 64, 256, 1, 1, 0, 1),     # layer1 c3: HBM-bound, K=64
    "L2": (8, 32, 52, 256, 256, 3, 1, 2, 0),      # layer3 3x3
    "L3": (8, 128, 208, 256, 256, 1, 1, 0, 0),    # up4 lateral 1x1
    "L4": (8, 32, 52, 256, 1024, 1, 1, 0, 1),     # layer3 c3
    "L5": (8, 128, 208, 64, 64, 3, 1, 1, 0),      # layer1 3x3
    "L6": (8, 16, 26, 512, 512, 3, 1, 2, 0),      # layer4 3x3
    "L7": (8, 64, 104, 128, 512, 1, 1, 0, 1),     # layer2 c3
    "L8": (8, 64, 104, 128, 128, 3, 1, 1, 0),     # layer2 3x3
    "L9": (8, 128, 208, 256, 43, 3, 1, 2, 0),     # last-stage keypoint/PAF head 3x3
    "L10": (8, 128, 208, 256, 14, 3, 1, 8, 0),    # last-stage head, Cout 14
    "L11": (8, 64, 104, 512, 512, 1, 1, 0, 0),    # up3 skip1
    "L12": (8, 64, 104, 512, 256, 1, 1, 0, 0),    # up3 lateral
    "L13": (8, 32, 52, 1024, 1024, 1, 1, 0, 0),   # up2 skip1
    "L14": (8, 32, 52, 1024, 256, 1, 1, 0, 0),    # layer3 c1
    "L15": (8, 64, 104, 256, 512, 1, 1, 0, 0),    # up3 skip2
    "L16": (8, 64, 104, 512, 128, 1, 1, 0, 0),    # layer2 c1
    "L17": (8, 16, 26, 2048, 2048, 1, 1, 0, 0),   # up1 skip1
    "L18": (8, 16, 26, 512, 2048, 1, 1, 0, 1),    # layer4 c3
    "L19": (8, 128, 208, 256, 64, 1, 1, 1, 0),    # layer1 c1 / cross_conv
    "L6b": (16, 16, 26, 512, 512, 3, 1, 2, 0),    # layer4 3x3 at twice the batch: what a 2-way split-K grid would look like to the CUs
    "L22": (8, 16, 26, 2048, 512, 1, 1, 2, 0),    # layer4 c1
    "L22b": (16, 16, 26, 2048, 512, 1, 1, 2, 0),
    "L18b": (16, 16, 26, 512, 2048, 1, 1, 0, 1),
    "L2b": (16, 32, 52, 256, 256, 3, 1, 2, 0),
    "L14b": (16, 32, 52, 1024, 256, 1, 1, 0, 0),
    "L20": (8, 128, 208, 256, 256, 1, 1, 0, 0, 1),  # up4.out: lateral + fused bilinear add of the 64x104 up_conv output
    "L21": (8, 64, 104, 512, 256, 1, 1, 0, 0, 1),   # up3.out
}


def build(B, H, W, Cin, Cout, k, s, tile, res, dev, x3=False, up=0):
    """x3: split-precision op (hi/lo planes: strides and weight matrix doubled, smap_op.precision = 1)."""
    lib = L.load()
    pl = 2 if x3 else 1
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    bn = TILES[tile][1]
    cout_pad = (Cout + bn - 1) // bn * bn
    K = k * k * Cin
    al = lambda n: (n + 255) // 256 * 256
    c8 = (Cout + 7) // 8 * 8
    x_b, o_b = al(B * H * W * Cin * 2 * pl), al(B * Ho * Wo * c8 * 2 * pl)
    u_b = al(B * (Ho // 2) * (Wo // 2) * c8 * 2 * pl) if up else 0
    arena = (torch.randn((16384 + x_b + 2 * o_b + u_b) // 2 + 128, device=dev) * 0.5).half()
    w_b = al(cout_pad * K * 2 * pl)
    blob = torch.zeros(w_b + al(cout_pad * 4), dtype=torch.uint8, device=dev)
    blob[:cout_pad * K * 2 * pl] = (torch.randn(cout_pad * K * pl, device=dev) * K ** -0.5).half().view(torch.uint8)
    op = L.SmapOp()
    op.kind, op.B, op.H, op.W, op.Cin, op.in_stride_c, op.in_c_off = 0, B, H, W, Cin, Cin * pl, 0
    op.Ho, op.Wo, op.Cout, op.ksize, op.stride, op.pad, op.relu = Ho, Wo, Cout, k, s, pad, 1
    op.cout_pad, op.out_stride_c, op.out_c_off, op.out_fp32, op.tile = cout_pad, c8 * pl, 0, 0, tile
    op.precision, op.acc_scale = int(x3), 1.0
    op.w_pairs = int(os.environ.get("SMAP_WPAIRS", "0"))          # random weights: only the address pattern differs
    op.in_off, op.out_off, op.w_off, op.bias_off = 16384, 16384 + x_b + o_b, 0, w_b
    op.res_off = 16384 + x_b if res else -1
    op.add1_off = op.add2_off = op.ext_off = -1
    for i in range(3):
        op.aux_off[i] = -1
    if up:
        op.aux_off[0], op.aux_h[0], op.aux_w[0] = 16384 + x_b + 2 * o_b, Ho // 2, Wo // 2
    h = C.c_void_p()
    L.check(lib.smap_plan_create(C.byref(op), 1, C.byref(h)), "create")
    flops = 2.0 * B * Ho * Wo * Cout * K
    byts = (B * H * W * Cin * 2 + B * Ho * Wo * Cout * 2 * (2 if res else 1) + cout_pad * K * 2) * pl
    return lib, h, arena, blob, flops, byts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    ap.add_argument("--tile-override", default="", help="e.g. L2:0,L6:5")
    ap.add_argument("--x3", action="store_true", help="split-precision instances (smap_op.precision = 1)")
    ap.add_argument("--rotate", type=int, default=1,
                    help="cycle through this many copies of the arena on ONE stream: with copies x tensor bytes > 256 MB every "
                         "launch finds its operands cold in the Infinity Cache, as inside the full schedule")
    ap.add_argument("--streams", type=int, default=1,
                    help="launch the iterations round-robin on this many streams (own arena each): the difference to one "
                         "stream is the drain-and-dispatch gap between dependent launches that a second stream can hide")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ov = dict(kv.split(":") for kv in args.tile_override.split(",") if ":" in kv)
    names = [n for n in PRESETS if not args.only or n in args.only.split(",")]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in names:
        p = list(PRESETS[n])
        if n in ov:
            p[7] = int(ov[n])
        lib, h, arena, blob, flops, byts = build(*p[:9], dev, x3=args.x3, up=p[9] if len(p) > 9 else 0)
        if args.streams > 1:
            streams = [torch.cuda.Stream(dev) for _ in range(args.streams)]
            arenas = [arena] + [arena.clone() for _ in range(args.streams - 1)]
            def run(i=[0]):
                k = i[0] % args.streams
                i[0] += 1
                L.check(lib.smap_plan_run(h, None, C.c_void_p(arenas[k].data_ptr()), C.c_void_p(blob.data_ptr()), None,
                                          C.c_void_p(streams[k].cuda_stream)), "run")
        elif args.rotate > 1:
            arenas = [arena] + [arena.clone() for _ in range(args.rotate - 1)]
            def run(i=[0]):
                k = i[0] % args.rotate
                i[0] += 1
                L.check(lib.smap_plan_run(h, None, C.c_void_p(arenas[k].data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
        else:
            run = lambda: L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()),
                                                    None, st), "run")
        for _ in range(3 * args.streams):
            run()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(args.iters):
            run()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / args.iters
        print(f"{n} {tuple(p)} tile={TILES[p[7]]}#{p[7]}: {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s  {byts / us / 1e3:7.0f} GB/s",
              flush=True)
        lib.smap_plan_destroy(h)


if __name__ == "__main__":
    main()
