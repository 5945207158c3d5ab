#!/usr/bin/env python3
"""Where does csrc/convb.hip go wrong, if it does?  Runs the single-op harness of tests/test_backbone_gpu.py in its partial
modes and prints the error by tile position / channel block / plane, so that ONE GPU visit names the phase and the index.

    python tools/debug/convb_probe.py [tile ...]          # default 90 91
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_backbone_gpu as T  # noqa: E402


def report(tile, mode, B=2, H=16, W=32, adds=False):
    th = 4 if tile in (90, 92) else 8
    try:
        got, ref = T._run_block(B, H, W, tile, adds, seed=7, mode=mode)
    except Exception as e:                                   # noqa: BLE001
        print(f"tile {tile} mode {mode}: EXCEPTION {e!r}")
        return
    err = (got - ref).abs()
    scale = ref.abs().max().item()
    bad = err > 3e-6 * scale + 1e-6
    print(f"tile {tile} {B}x{H}x{W} mode {mode:10s} adds {int(adds)}: max err {err.max().item():.3e} (scale {scale:.3f}) "
          f"bad {int(bad.sum())}/{bad.numel()} finite {bool(torch.isfinite(got).all())}")
    if not bad.any():
        return
    e = err.numpy()
    C = e.shape[-1]                                          # 256, or 512 for csrc/convc.hip (tile 94)
    by_row = e.reshape(B, H // th if H % th == 0 else -1, th, W, C).max((0, 1, 3, 4)) if H % th == 0 else None
    by_col = e.reshape(B, H, W // 16, 16, C).max((0, 1, 2, 4)) if W % 16 == 0 else None
    by_ch32 = e.reshape(B, H, W, C // 32, 32).max((0, 1, 2, 4))
    by_ch8 = e.reshape(B, H, W, C // 8, 8).max((0, 1, 2, 4))
    print("   max err by row inside the tile :", None if by_row is None else np.array2string(by_row, precision=2))
    print("   max err by col inside the tile :", None if by_col is None else np.array2string(by_col, precision=2))
    print("   max err by 32-channel block    :", np.array2string(by_ch32, precision=2))
    print("   max err by 8-channel granule   :", np.array2string(by_ch8, precision=2, max_line_width=200))
    idx = np.argwhere(bad.numpy())[:6]
    for b, y, x, c in idx:
        print(f"   [{b},{y},{x},{c}] got {got[b, y, x, c].item():.6f} ref {ref[b, y, x, c].item():.6f}")


if __name__ == "__main__":
    tiles = [int(t) for t in sys.argv[1:]] or [90, 91, 92, 93]
    for tile in tiles:
        for mode in ("residual", "no_c1", "centre_tap", "full"):
            report(tile, mode)
        report(tile, "full", B=1, H=13, W=52, adds=tile < 92 or tile == 94)
