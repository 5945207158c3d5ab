#!/usr/bin/env python3
"""Where a persistent conv workgroup (csrc/convp.hip) spends its life: summed phase times of compute wave 0 and loader
wave 0 of every workgroup (diagnostics build: tools/build_ablate.py --trace).
    SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_trace.so python tools/trace_convp.py L3:60 L13:60 [--f16]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from smap_amd import lib as L  # noqa: E402
from bench_conv import PRESETS, build  # noqa: E402


def main():
    x3 = "--f16" not in sys.argv
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in names:
        name, _, tile = n.partition(":")
        p = list(PRESETS[name])
        p[7] = int(tile)
        lib, h, arena, blob, flops, byts = build(*p, dev, x3=x3)
        arenas = [arena, arena.clone(), arena.clone()]
        trace = torch.zeros((1024, 16), dtype=torch.int64, device=dev)
        os.environ["SMAP_TRACE_PTR"] = str(trace.data_ptr())
        run = lambda k: L.check(lib.smap_plan_run(h, None, C.c_void_p(arenas[k % 3].data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
        for k in range(4):
            run(k)
        torch.cuda.synchronize()
        trace.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(4); e1.record()
        torch.cuda.synchronize()
        t = trace.cpu().numpy().astype(np.float64)
        t = t[t[:, 0] != 0]
        tick = 0.01            # us per s_memtime tick (100 MHz)
        f = lambda a: f"{np.mean(a):8.2f} (p10 {np.percentile(a, 10):7.2f} p90 {np.percentile(a, 90):7.2f})"
        life_c, life_l = (t[:, 1] - t[:, 0]) * tick, (t[:, 9] - t[:, 8]) * tick
        tiles, ktiles = t[:, 5], t[:, 13]
        print(f"{n} {tuple(p)}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {len(t)} workgroups, tiles/wg {tiles.mean():.2f}, K tiles/wg {ktiles.mean():.1f}")
        print(f"   compute wave 0: life {f(life_c)}\n      barrier wait {f(t[:, 2] * tick)}\n      ds_read+MFMA {f(t[:, 3] * tick)}\n      epilogue     {f(t[:, 4] * tick)}")
        print(f"      per K tile: barrier {np.mean(t[:, 2] / ktiles) * tick:.3f} us, ds_read+MFMA {np.mean(t[:, 3] / ktiles) * tick:.3f} us; epilogue per tile {np.mean(t[:, 4] / tiles) * tick:.2f} us")
        print(f"   loader wave 0 : life {f(life_l)}\n      vmcnt wait   {f(t[:, 10] * tick)}\n      barrier wait {f(t[:, 11] * tick)}\n      issue        {f(t[:, 12] * tick)}")
        print(f"      per K tile: vmcnt {np.mean(t[:, 10] / ktiles) * tick:.3f} us, barrier {np.mean(t[:, 11] / ktiles) * tick:.3f} us, issue {np.mean(t[:, 12] / ktiles) * tick:.3f} us")
        start = (t[:, 0] - t[:, 0].min()) * tick
        print(f"   workgroup start spread: p50 {np.percentile(start, 50):.1f} us, max {start.max():.1f} us; end spread {((t[:, 1] - t[:, 1].min()) * tick).max():.1f} us")
        lib.smap_plan_destroy(h)


if __name__ == "__main__":
    main()
