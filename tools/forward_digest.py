#!/usr/bin/env python3
"""sha256 of every output of one seeded SMAP forward (random-init weights of seed 0, inputs of seed 1234) through the library that
SMAP_HIP_LIB names (default: the shipped one): two builds that claim to compute the same thing bit for bit print the same lines.
    python tools/forward_digest.py [--batch 8] [--precision x3|f16]"""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--precision", default="x3")
args = ap.parse_args()
from benchkit.workload import make_cfg  # noqa: E402
from model.smap import SMAP  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval().to(dev)
net.precision = args.precision
eng = net.engine(args.batch, 512, 832, dev)
imgs = torch.randn(args.batch, 3, 512, 832, generator=torch.Generator().manual_seed(1234)).to(dev)
out = eng.new_output()
eng.run(imgs, out=out)
torch.cuda.synchronize()
views = out if isinstance(out, (list, tuple)) else [out]
for i, t in enumerate(views):
    t = t if torch.is_tensor(t) else torch.as_tensor(t)
    print("output", i, tuple(t.shape), str(t.dtype), hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest())
