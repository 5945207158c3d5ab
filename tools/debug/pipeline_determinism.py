"""Full-size determinism probe of the multi-stream pipeline: the SAME batch submitted N times back to back must give N identical
record lists, at any depth (two backbones + the post stream really overlap at 512x832; the 64x96 test cannot show a race that
needs that).  python tools/debug/pipeline_determinism.py [B] [flip] [depth] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from exps.stage3_root2.config import cfg
from smap_amd.pipeline import PosePipeline, make_pipeline

B, flip, depth, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = "cuda:0"
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval()
sd = recipe_state_dict(net.state_dict())
for k in list(sd):
    if k.endswith("up4.res_conv2.bn.bias"):
        sd[k] = sd[k] + 40.0
net.load_state_dict(sd)
net = net.to(dev)
g = torch.Generator().manual_seed(3)
imgs = torch.randn(B, 3, 512, 832, generator=g).to(dev)
cams = np.tile(np.array([0.5, 832, 512, 416, 256, 832, 832, 416, 256], np.float64), (B, 1))
LF = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # frames per launch (coalesced pipeline) or 0
pipe = make_pipeline(net, cfg, B, 512, 832, dev, launch_frames=LF, depth=depth, do_flip=bool(flip))
allr = []
for i in range(N):
    allr += pipe.submit(imgs, cams, [f"f{j}" for j in range(B)]) or []
allr += pipe.flush() or []
names = [r["image_path"] for r in allr]
per = names.index(names[0], 1) if names.count(names[0]) > 1 else len(names)      # records per submitted batch
outs = [allr[i:i + per] for i in range(0, len(allr), per)]
print("pipeline:", type(pipe).__name__, "frames per launch", pipe.frames_per_launch)
key = lambda recs: [(r["image_path"], np.asarray(r["pred_2d"]).tobytes(), np.asarray(r["pred_3d"]).tobytes(), np.asarray(r["root_d"]).tobytes()) for r in recs]
ref = key(outs[0])
bad = [i for i, o in enumerate(outs) if key(o) != ref]
print(f"B={B} flip={flip} depth={depth}: {len(outs)} batches, {per} records each, persons {sum(len(r['root_d']) for r in outs[0])}, differing batches: {bad}")
for i in bad[:3]:
    for a, b in zip(outs[0], outs[i]):
        if len(a["root_d"]) != len(b["root_d"]):
            print("  batch", i, a["image_path"], "persons", len(a["root_d"]), "vs", len(b["root_d"]))
        else:
            d = np.abs(np.asarray(a["pred_3d"]) - np.asarray(b["pred_3d"])).max() if len(a["root_d"]) else 0
            if d:
                print("  batch", i, a["image_path"], "max |d pred_3d|", d)
