"""Which co-running kernel upsets the head sum?  Engine A repeats its HEADSUM ops; engine B repeats ONE op of its schedule (one
representative per tile id / op kind) next to it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from smap_amd.engine import BackboneEngine, OP_HEADSUM, OP_CONV

B, REP = int(sys.argv[1]), int(sys.argv[2])
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
ref = eng.out.clone()
n = eng.n_ops
nh = sum(1 for op in eng.graph.ops if op.kind == OP_HEADSUM)
reps = {}
for i, op in enumerate(eng.graph.ops):
    key = ("conv", op.p["tile"], bool(op.aux), op.p["ksize"]) if op.kind == OP_CONV else ("kind", op.kind)
    reps.setdefault(key, []).append(i)
for key, idxs in sorted(reps.items(), key=lambda kv: str(kv[0])):
    i = idxs[len(idxs) // 2]
    op = eng.graph.ops[i]
    bad = 0
    for it in range(REP):
        with torch.cuda.stream(s1):
            for _ in range(12):
                sib.run(imgs, first=i, count=1)
        with torch.cuda.stream(s0):
            for _ in range(6):
                eng.run(imgs, first=n - nh, count=nh)
        torch.cuda.synchronize()
        bad += not torch.equal(eng.out, ref)
    desc = f"{op.out.name if op.out is not None else 'headsum'}"
    print(f"{str(key):36} op {i:3d} {desc[-40:]:40} ({len(idxs)} such ops): head sum wrong in {bad} of {REP}")
