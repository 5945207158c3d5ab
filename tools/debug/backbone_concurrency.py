"""Two executors of one schedule on two streams, launched back to back N times: every output must equal the serial result bit for
bit.  python tools/debug/backbone_concurrency.py [B] [flip] [N] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from exps.stage3_root2.config import cfg

B, flip, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cuda:0"
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval()
if len(sys.argv) > 4:
    net.precision = sys.argv[4]
sd = recipe_state_dict(net.state_dict())
net.load_state_dict(sd)
net = net.to(dev)
kpt = cfg.DATASET.KEYPOINT.NUM
fp = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [kpt + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
eng = net.engine(B, 512, 832, dev, flip_pair=fp if flip else None)
engs = [eng, eng.sibling()]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
imgs = torch.randn(B, 3, 512, 832, generator=torch.Generator().manual_seed(3)).to(dev)
ref = eng.new_output()
eng.run(imgs, out=ref)
torch.cuda.synchronize()
outs = [eng.new_output() for _ in range(N)]
for i in range(N):
    with torch.cuda.stream(streams[i % 2]):
        engs[i % 2].run(imgs, out=outs[i])
torch.cuda.synchronize()
bad = [i for i in range(N) if not torch.equal(outs[i], ref)]
print(f"backbone B={B} flip={flip} prec={net.precision}: {N} concurrent runs, differing: {bad}")
for i in bad[:4]:
    d = (outs[i] - ref).abs()
    nz = d.nonzero().flatten()
    hms_end = B * 43 * 128 * 208
    print(f"   run {i}: {nz.numel()} floats differ, max {d.max().item():.3g}, first idx {nz[0].item()} last {nz[-1].item()} (hms block ends at {hms_end})")
    if "--tensors" in sys.argv:
        pass
