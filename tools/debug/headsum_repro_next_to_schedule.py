"""Bisecting EXPERIMENTS R3.6: the three variants of tools/experiments/headsum_runtime_index_repro.hip (0 = the old kernel, 1 = run-time
index into the argument struct but no private array, 2 = fixed) run next to the library's REAL schedule on a second stream."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from smap_amd.engine import BackboneEngine

lib = C.CDLL(os.path.join(ROOT, "tools", "experiments", "headsum_runtime_index_repro.so"))
lib.repro_headsum.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
B, N = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval()
eng = BackboneEngine(recipe_state_dict(net.state_dict()), B, 512, 832, dev, precision="x3")
imgs = torch.randn(B, 3, 512, 832, generator=torch.Generator().manual_seed(3)).to(dev)
g = torch.Generator().manual_seed(5)
src = [torch.randn(B, h, w, 48, generator=g).to(dev) for h, w in ((128, 208), (64, 104), (32, 52))]
out = torch.zeros(B, 43, 128, 208, device=dev)
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def run(V, stream):
    rc = lib.repro_headsum(V, *[C.c_void_p(t.data_ptr()) for t in src], C.c_void_p(out.data_ptr()), B, C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc
run(2, s0)
torch.cuda.synchronize()
ref = out.clone()
for overlap in (False, True):
    res = {}
    for V in (0, 1, 2):
        bad = 0
        for it in range(N):
            if overlap:
                with torch.cuda.stream(s1):
                    eng.run(imgs)
            for _ in range(6):
                run(V, s0)
            torch.cuda.synchronize()
            bad += not torch.equal(out, ref)
        res[V] = bad
    print("next to a running schedule:" if overlap else "alone:", {f"V{v}": f"{b} of {N}" for v, b in res.items()})
