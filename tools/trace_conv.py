#!/usr/bin/env python3
"""Per-workgroup phase timeline of the conv kernel (diagnostics build, tools/build_ablate.py --trace):
    SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_trace.so [SMAP_TRACE_X3=1] python tools/trace_conv.py [L1 L3:20 ...]
Stamps (s_memtime, 100 MHz constant clock on gfx950 -> printed in us): start, setup done, first K
tile landed, K loop done, LDS staging done, stores retired; plus the summed wait at the top of each
K iteration and the CU the workgroup ran on."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from smap_amd import lib as L  # noqa: E402
from smap_amd.engine import TILES  # noqa: E402
from bench_conv import PRESETS, build  # noqa: E402


def main():
    names = sys.argv[1:] or ["L1", "L3", "L2", "L4"]
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in names:
        name, _, tile = n.partition(":")
        p = list(PRESETS[name])
        if tile:
            p[7] = int(tile)
        lib, h, arena, blob, flops, byts = build(*p, dev, x3=bool(os.environ.get('SMAP_TRACE_X3')))
        B, H, W, Cin, Cout, k, s = p[:7]
        bm, bn = TILES[p[7]]
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        nblk = -(-(B * Ho * Wo) // bm) * (-(-Cout // bn))
        trace = torch.zeros((max(nblk, 16384), 8), dtype=torch.int64, device=dev)      # halo tiles: 2-D pixel tiles, more blocks
        os.environ["SMAP_TRACE_PTR"] = str(trace.data_ptr())
        run = lambda: L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        t = trace.cpu().numpy()
        t = t[t[:, 0] != 0]
        nblk = len(t)
        vmw, barw = (t[:, 6] & 0xffffffff).astype(np.float64), (t[:, 6] >> 32).astype(np.float64)
        t = t.astype(np.float64)
        if p[7] >= 30:
            t[:, 6] = vmw + barw
        tick = 0.01                                   # us per s_memtime tick (100 MHz)
        t0 = t[:, 0].min()
        start = (t[:, 0] - t0) * tick
        setup = (t[:, 1] - t[:, 0]) * tick
        first = (t[:, 2] - t[:, 1]) * tick
        loop = (t[:, 3] - t[:, 2]) * tick
        stg = (t[:, 4] - t[:, 3]) * tick
        epi = (t[:, 5] - t[:, 4]) * tick
        life = (t[:, 5] - t[:, 0]) * tick
        wait = t[:, 6] * tick
        span = (t[:, 5].max() - t0) * tick
        cu = (t[:, 7].astype(np.int64) >> 8) & 0xF
        print(f"{n} {tuple(p)} tile {bm}x{bn}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, stamp span {span:.1f} us, blocks {nblk}")
        f = lambda a: f"{np.mean(a):7.2f} (p10 {np.percentile(a, 10):6.2f} p90 {np.percentile(a, 90):6.2f})"
        print(f"   setup {f(setup)}\n   first tile {f(first)}\n   K loop rest {f(loop)}  of which waits(all iters) {f(wait)}\n   staging {f(stg)}\n   epilogue+drain {f(epi)}\n   lifetime {f(life)}")
        if p[7] >= 30:
            print(f"   waits split: vmcnt {f(vmw * tick)}  barrier {f(barw * tick)}")
        hist, edges = np.histogram(start, bins=8)
        print("   block start times (us):", " ".join(f"{int(c)}@{e:.0f}" for c, e in zip(hist, edges)))
        lib.smap_plan_destroy(h)


if __name__ == "__main__":
    main()
