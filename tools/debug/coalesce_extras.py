"""Coalesced vs per-batch pipeline with the bench's `extra` maps (synthetic scenes, a different one per step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from exps.stage3_root2.config import cfg
from smap_amd.pipeline import make_pipeline
from benchkit.workload import synth_scene

B, dev = 8, "cuda:0"
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval()
sd = recipe_state_dict(net.state_dict())
for k in list(sd):
    if k.endswith("up4.res_conv2.bn.bias"):
        sd[k] = sd[k] + 40.0
net.load_state_dict(sd)
net = net.to(dev)
imgs = torch.randn(B, 3, 512, 832, generator=torch.Generator().manual_seed(3)).to(dev)
cams = np.tile(np.array([0.5, 832, 512, 416, 256, 832, 832, 416, 256], np.float64), (B, 1))
KS = (0, 2, 8, 20)
synth = {}
for K in KS:
    sc = [synth_scene(K, seed=10 * K + i)[:2] for i in range(B)]
    synth[K] = (torch.from_numpy(np.stack([s_[0] for s_ in sc])).to(dev), torch.from_numpy(np.stack([s_[1] for s_ in sc])).to(dev))
res = {}
for lf in (0, 16):
    pipe = make_pipeline(net, cfg, B, 512, 832, dev, launch_frames=lf, n_extra=1, depth=2, numpy_records=True)
    recs = []
    for i in range(8):
        K = KS[i % 4]
        recs += pipe.submit(imgs, cams, [f"f{j}" for j in range(B)], extra=[(f"synthK{K}", synth[K][0], synth[K][1], None)]) or []
    recs += pipe.flush() or []
    res[lf] = recs
    summary = {}
    for r in recs:
        summary.setdefault(r["image_path"], []).append(len(r["root_d"]))
    print("launch_frames", lf, type(pipe).__name__, "records", len(recs))
    for k in sorted(summary)[:48]:
        print("   ", k, summary[k])
