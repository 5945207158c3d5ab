"""Do plain torch copy kernels see engine A's (static) fp32 head tensors change while engine B runs schedules next to them?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from smap_amd.engine import BackboneEngine, OP_HEADSUM

B, N = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval()
sd = recipe_state_dict(net.state_dict())
eng = BackboneEngine(sd, B, 512, 832, dev, precision="x3")
sib = eng.sibling()
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
imgs = torch.randn(B, 3, 512, 832, generator=torch.Generator().manual_seed(3)).to(dev)
eng.run(imgs); sib.run(imgs)
torch.cuda.synchronize()
heads = [t for op in eng.graph.ops if op.kind == OP_HEADSUM for t in op.aux]
lo, hi = min(t.off for t in heads), max(t.off + t.nbytes for t in heads)
region = eng.arena[lo:hi]
ref = region.clone()
torch.cuda.synchronize()
print("head tensors:", [(t.name, t.off, t.nbytes) for t in heads], "region bytes", hi - lo)
bad = 0
for it in range(N):
    with torch.cuda.stream(s1):
        sib.run(imgs)
    with torch.cuda.stream(s0):
        copies = [region.clone() for _ in range(6)]
    torch.cuda.synchronize()
    for j, c in enumerate(copies):
        if not torch.equal(c, ref):
            d = (c != ref).nonzero().flatten()
            bad += 1
            print(f"iter {it} copy {j}: {d.numel()} bytes differ, first offsets {d[:12].tolist()}")
print("copies that differed:", bad, "of", N * 6)
