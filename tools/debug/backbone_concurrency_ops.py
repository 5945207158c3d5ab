"""Which op goes wrong when two executors of one schedule overlap?  Engines WITHOUT arena reuse (every tensor keeps its bytes), the
arena of each concurrent run compared with the serial run's, differing bytes mapped back to tensors / ops.
python tools/debug/backbone_concurrency_ops.py [B] [N] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from helpers import make_cfg
from recipe import recipe_state_dict
from model.smap import SMAP
from smap_amd.engine import BackboneEngine, TILES

B, N = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith('--') else "x3"
dev = "cuda:0"
torch.manual_seed(0)
net = SMAP(make_cfg((128, 208))).eval()
sd = recipe_state_dict(net.state_dict())
flip = "--flip" in sys.argv
fp = None
if flip:
    from exps.stage3_root2.config import cfg
    fp = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [cfg.DATASET.KEYPOINT.NUM + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
reuse = "--reuse" in sys.argv
eng = BackboneEngine(sd, B, 512, 832, dev, reuse=reuse, precision=prec, flip_pair=fp)
engs = [eng, eng.sibling()]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
imgs = torch.randn(B, 3, 512, 832, generator=torch.Generator().manual_seed(3)).to(dev)
eng.run(imgs)
torch.cuda.synchronize()
ref = eng.arena.clone()
ref_out = eng.out.clone()
n_out_bad = 0
g = eng.graph
spans = sorted((t.off, t.off + t.nbytes, t.name) for t in g.tensors)
producer = {op.out.name: (i, op) for i, op in enumerate(g.ops) if op.out is not None}
seen = {}
for it in range(N):
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            engs[k].run(imgs)
    torch.cuda.synchronize()
    for k in range(2):
        a = engs[k].arena
        if not torch.equal(engs[k].out, ref_out):
            d = (engs[k].out != ref_out).nonzero().flatten().cpu().tolist()
            hw = 128 * 208
            n_out_bad += 1
            print(f"iter {it} engine {k}: OUT differs at {len(d)} floats; arena equal: {torch.equal(a[16384:], ref[16384:])}; "
                  f"(frame, channel, y, x) of the first: {[(i // (43 * hw), i // hw % 43, i % hw // 208, i % 208) for i in d[:16]]}")
        if torch.equal(a[16384:], ref[16384:]):
            continue
        diff = (a != ref).nonzero().flatten()
        diff = diff[diff >= 16384]
        offs = diff.cpu().tolist()
        hit = {}
        for o in offs:
            for lo, hi, name in spans:
                if lo <= o < hi:
                    hit.setdefault(name, []).append(o - lo)
                    break
        first = min(hit, key=lambda n: producer[n][0])
        i, op = producer[first]
        p = op.p
        desc = f"op {i} kind {op.kind} {first}"
        if op.kind == 0:
            desc += f" tile {p['tile']} Cin {p['Cin']} Cout {p['Cout']} k{p['ksize']} s{p['stride']} in {op.inp.H}x{op.inp.W}"
        nb = len(hit[first])
        t = next(t for t in g.tensors if t.name == first)
        px = sorted({o // (t.C * t.esize * t.planes) for o in hit[first]})
        print(f"iter {it} engine {k}: {len(offs)} bytes differ in {len(hit)} tensors; FIRST {desc}: {nb} bytes, pixels {px[:6]}{'...' if len(px) > 6 else ''} (W {t.W}, rows {sorted({q // t.W % t.H for q in px})[:6]})")
        seen[desc] = seen.get(desc, 0) + 1
print("summary:", seen if seen else "no arena differences", "| runs with a differing OUT buffer:", n_out_bad)
