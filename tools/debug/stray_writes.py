"""Does a schedule write outside its own buffers?  Engine A idle (arena / out snapshotted), guard tensors around the allocations,
engine B runs N schedules; anything that changed was written by B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from smap_amd.engine import BackboneEngine

B, N = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval()
sd = recipe_state_dict(net.state_dict())
guard = lambda: torch.full((64 << 20,), 0x5A, dtype=torch.uint8, device=dev)
g1 = guard()
eng = BackboneEngine(sd, B, 512, 832, dev, precision="x3")
g2 = guard()
sib = eng.sibling()
g3 = guard()
imgs = torch.randn(B, 3, 512, 832, generator=torch.Generator().manual_seed(3)).to(dev)
eng.run(imgs); sib.run(imgs)
torch.cuda.synchronize()
snap_arena, snap_out = eng.arena.clone(), eng.out.clone()
w_snap = eng.weights.clone()
print("addresses: g1 %x eng.arena %x eng.out %x g2 %x sib.arena %x sib.out %x g3 %x weights %x" % (
    g1.data_ptr(), eng.arena.data_ptr(), eng.out.data_ptr(), g2.data_ptr(), sib.arena.data_ptr(), sib.out.data_ptr(), g3.data_ptr(), eng.weights.data_ptr()))
for it in range(N):
    sib.run(imgs)
torch.cuda.synchronize()
for name, t, ref in (("engine A arena", eng.arena, snap_arena), ("engine A out", eng.out, snap_out), ("weights", eng.weights, w_snap)):
    d = (t != ref).nonzero().flatten()
    print(name, "changed elements:", d.numel(), d[:8].tolist())
for name, t in (("g1", g1), ("g2", g2), ("g3", g3)):
    d = (t != 0x5A).nonzero().flatten()
    print(name, "changed bytes:", d.numel(), d[:8].tolist())
