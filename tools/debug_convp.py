#!/usr/bin/env python3
"""Debug aid: one small conv through a convp tile; where do the values land?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from test_backbone_gpu import _run_single_conv

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 60
x3 = (sys.argv[2] == "x3") if len(sys.argv) > 2 else True
got, ref, cout = _run_single_conv(2, 16, 24, 64, 256, 1, 1, tile, False, False, False, seed=1, x3=x3)
g = got.reshape(-1, got.shape[-1])[:, :cout]
r = ref.reshape(-1, cout)
err = (g - r).abs()
print("max err", err.max().item(), "ref max", r.abs().max().item(), "frac wrong", (err > 1e-2).float().mean().item())
# per-pixel: which reference (pixel, channel) does got[p, c] equal?
M = g.shape[0]
for p in (0, 1, 33, 70, 200):
    row = []
    for c in list(range(0, 20)) + [32, 40, 64, 65, 128, 255]:
        v = g[p, c]
        d = (r - v).abs()
        idx = d.argmin().item()
        pp, cc = idx // cout, idx % cout
        row.append(f"{c}->({pp},{cc},{d.min().item():.1e})")
    print("pixel", p, " ".join(row))
wrong_px = (err > 1e-2).any(1).nonzero().flatten()
wrong_ch = (err > 1e-2).any(0).nonzero().flatten()
print("wrong pixels:", wrong_px[:40].tolist(), "n =", len(wrong_px), "of", M)
print("wrong channels:", wrong_ch[:64].tolist(), "n =", len(wrong_ch))
