#!/usr/bin/env python3
"""Where a workgroup of csrc/convb.hip spends its life (diagnostics build: python tools/build_ablate.py --trace):

    SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_trace.so python tools/trace_convb.py [tile ...]      # default 90 91

One whole-Bottleneck op at the layer1 size (8 x 128 x 208, C = 256), stamps of wave 0 of every workgroup on the 100 MHz
s_memtime clock: set-up | phase 1 (c1 on the halo, x from HBM) | y1 write + first weight slot | phase 2 (3x3) | y2 write +
phase 3 (tail + epilogue) | store drain; plus how the workgroups pack onto the CUs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_backbone_gpu as T  # noqa: E402


def main():
    tiles = [int(t) for t in sys.argv[1:]] or [90, 91]
    B, H, W = 8, 128, 208
    for tile in tiles:
        trace = torch.zeros((16384, 8), dtype=torch.int64, device="cuda:0")
        os.environ["SMAP_TRACE_PTR"] = str(trace.data_ptr())
        for rep in range(3):
            trace.zero_()
            torch.cuda.synchronize()
            T._run_block(B, H, W, tile, False, seed=1, check=False)
        t = trace.cpu().numpy()
        t = t[t[:, 0] != 0].astype(np.float64)
        if not len(t):
            print(f"tile {tile}: no stamps (is SMAP_HIP_LIB the trace build?)")
            continue
        tick = 0.01
        d = np.diff(t[:, :7], axis=1) * tick
        life = (t[:, 6] - t[:, 0]) * tick
        span = (t[:, 6].max() - t[:, 0].min()) * tick
        names = ["set-up", "phase 1 (c1, x from HBM)", "y1 write + slot 0", "phase 2 (3x3)", "y2 + phase 3 (tail, epilogue)", "store drain"]
        print(f"tile {tile}: {len(t)} workgroups, launch span {span:.1f} us, workgroup life mean {life.mean():.1f} us (p10 {np.percentile(life, 10):.1f}, p90 {np.percentile(life, 90):.1f})")
        for k, n in enumerate(names):
            print(f"   {n:32s} mean {d[:, k].mean():7.2f} us   p10 {np.percentile(d[:, k], 10):7.2f}   p90 {np.percentile(d[:, k], 90):7.2f}")
        hw = t[:, 7].astype(np.int64)
        cu = (hw >> 8) & 0xf
        se = (hw >> 13) & 0x7
        start = (t[:, 0] - t[:, 0].min()) * tick
        end = (t[:, 6] - t[:, 0].min()) * tick
        grid = np.arange(0, span, 1.0)
        resident = ((start[None, :] <= grid[:, None]) & (end[None, :] > grid[:, None])).sum(1)
        print(f"   workgroups resident at once: mean {resident.mean():.0f}, max {resident.max()} (256 CUs x 2 = 512)")


if __name__ == "__main__":
    main()
