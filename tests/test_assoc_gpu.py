"""GPU parity: HIP association / lifting kernels (through the C ABI via smap_amd.dapalib)
against the CPU oracle -- bit-exact for every integer AND float of nms / paf / group
(both sides are compiled without FMA contraction) -- and against the reference goldens."""
import numpy as np
import pytest
import torch

from helpers import synth_scene, noise_scene
from fixture_maps import expand
from recipe import recipe_state_dict
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def scenes():
    out = [synth_scene(k, seed=100 + k) for k in (0, 1, 2, 8, 20)]
    out += [synth_scene(8, seed=7, noise=0.05, drop=0.3)[:2] + (None, None)]
    out += [noise_scene(1) + (None, None), noise_scene(2, amp=0.5) + (None, None)]
    out += [(np.zeros((43, 128, 208), np.float32), np.ones((128, 208), np.float32), None, None)]
    return [(h, r) for h, r, *_ in out]


@pytest.fixture(scope="module")
def batch():
    sc = scenes()
    hms = torch.from_numpy(np.stack([h for h, _ in sc])).to(DEV)
    rd = torch.from_numpy(np.stack([r for _, r in sc])).to(DEV)
    return sc, hms, rd


def test_hip_library_is_loaded():
    import smap_amd.lib as L
    L.load()
    assert "libsmap_hip.so" in open("/proc/self/maps").read()


def test_nms_paf_group_bit_exact(batch):
    import dapalib
    sc, hms, rd = batch
    bodys, counts, peaks, scores = dapalib.connect_batch(hms, rd, return_intermediate=True)
    torch.cuda.synchronize()
    peaks, scores, bodys, counts = peaks.cpu().numpy(), scores.cpu().numpy(), bodys.cpu().numpy(), counts.cpu().numpy()
    n_people = []
    for i, (h, r) in enumerate(sc):
        ob, opk, osc = O.connect(h, r)
        assert np.array_equal(bits(peaks[i]), bits(opk)), f"scene {i}: peaks differ"
        assert np.array_equal(bits(scores[i]), bits(osc)), f"scene {i}: paf scores differ"
        assert counts[i] == len(ob)
        assert np.array_equal(bits(bodys[i, :len(ob)]), bits(ob)), f"scene {i}: limb assignment differs"
        assert not bodys[i, len(ob):].any()
        n_people.append(len(ob))
    assert max(n_people) == 127 and 0 in n_people and 20 in n_people


def test_dist_flag_and_root_idx(batch):
    import dapalib
    sc, hms, rd = batch
    for root, dist in ((2, False), (0, True)):
        bodys, counts = dapalib.connect_batch(hms[:5], rd[:5], rootIdx=root, distFlag=dist)
        for i in range(5):
            pk = O.nms(sc[i][0])
            ob = O.group(pk, O.paf_score(sc[i][0], pk), sc[i][1], root, dist)
            assert int(counts[i]) == len(ob)
            assert np.array_equal(bits(bodys[i, :len(ob)].cpu().numpy()), bits(ob))


def test_reference_api_connect_extract(batch):
    import dapalib
    sc, hms, rd = batch
    for i in (2, 4, 8):
        h, r = sc[i]
        ob, opk, osc = O.connect(h, r)
        out = dapalib.connect(hms[i], rd[i].cpu(), 2, distFlag=True)        # rDepth on the host, like test.py:113
        assert out.device.type == "cpu" and out.dtype == torch.float32
        if len(ob) == 0:
            assert tuple(out.shape) == (0,)
        else:
            assert tuple(out.shape) == (len(ob), 15, 4) and np.array_equal(bits(out.numpy()), bits(ob))
        cands, pafs = dapalib.extract(hms[i])
        assert len(cands) == 15 and len(pafs) == 14
        for j in range(15):
            n = int(opk[j, 0, 0])
            assert tuple(cands[j].shape) == (n, 3) and np.array_equal(bits(cands[j].numpy()), bits(opk[j, 1:1 + n]))


def test_two_launch_peak_search_equals_the_fused_kernel_bit_for_bit(batch):
    """smap_nms_ws (mask of the whole batch with one thread per pixel, then scan + centroids: what extract_batch runs) against the
    single-launch smap_nms: the same [B,15,128,3] peak table bit for bit -- on the synthetic scenes (0..20 persons, the 127-peak cap in
    the noise scenes), on map sizes that are no multiple of a wave / of the mask kernel's workgroup, and through the raw C ABI incl. its
    argument checks (workspace too small / misaligned / missing)."""
    import ctypes as C
    import dapalib
    from smap_amd import lib as L
    _, hms, rd = batch
    pk_ws, sc_ws = dapalib.extract_batch(hms)
    pk_f, sc_f = dapalib.extract_batch(hms, fused_nms=True)
    assert torch.equal(pk_ws.view(torch.int32), pk_f.view(torch.int32)) and torch.equal(sc_ws.view(torch.int32), sc_f.view(torch.int32))
    assert pk_ws[:, :, 0, 0].max() >= 20
    for (H, W), seed in (((13, 21), 1), ((40, 52), 2), ((3, 3), 3), ((100, 300), 4), ((128, 208), 5)):
        rng = np.random.default_rng(seed)
        x = torch.from_numpy(rng.uniform(0, 1, (3, 43, H, W)).astype(np.float32)).to(DEV)
        a, _ = dapalib.extract_batch(x)
        b, _ = dapalib.extract_batch(x, fused_nms=True)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (H, W)
    lib = L.load()
    x = hms[:2].contiguous()
    nb = lib.smap_nms_workspace_bytes(2, 128, 208)
    assert nb == 2 * 15 * 416 * 8 and lib.smap_nms_workspace_bytes(0, 128, 208) == 0
    ws = torch.zeros(nb // 8 + 1, dtype=torch.int64, device=DEV)
    pk = torch.empty((2, 15, 128, 3), dtype=torch.float32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.smap_nms_ws(p(x), 2, 43, 128, 208, 0.2, p(pk), p(ws), nb, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(pk.view(torch.int32), pk_f[:2].view(torch.int32))
    assert lib.smap_nms_ws(p(x), 2, 43, 128, 208, 0.2, p(pk), p(ws), nb - 8, st) == -1                      # too small
    assert lib.smap_nms_ws(p(x), 2, 43, 128, 208, 0.2, p(pk), C.c_void_p(ws.data_ptr() + 4), nb, st) == -1   # misaligned
    assert lib.smap_nms_ws(p(x), 2, 43, 128, 208, 0.2, p(pk), None, nb, st) == -1


def test_other_map_sizes():
    import dapalib
    for (H, W), seed in (((16, 24), 1), ((32, 52), 2), ((64, 104), 3), ((100, 300), 4)):
        rng = np.random.default_rng(seed)
        hms = rng.uniform(0, 1, (2, 43, H, W)).astype(np.float32)
        hms[:, 15:] = rng.uniform(-1, 1, (2, 28, H, W))
        rd = rng.uniform(0.2, 1, (2, H, W)).astype(np.float32)
        b, c, pk, sc = dapalib.connect_batch(torch.from_numpy(hms).to(DEV), torch.from_numpy(rd).to(DEV),
                                             return_intermediate=True)
        for i in range(2):
            ob, opk, osc = O.connect(hms[i], rd[i])
            assert np.array_equal(bits(pk[i].cpu().numpy()), bits(opk))
            assert np.array_equal(bits(sc[i].cpu().numpy()), bits(osc))
            assert np.array_equal(bits(b[i, :len(ob)].cpu().numpy()), bits(ob))


@pytest.mark.parametrize("flip", [False, True], ids=["plain", "flip-tta"])
@pytest.mark.parametrize("precision", ["x3", "f16"])
def test_head_sum_writes_the_scaled_maps_bit_for_bit(flip, precision):
    """smap_op.scale_hms: the schedule's head sum stores hms / 255 | / 127 itself (test.py:111-112 fused into the backbone) -- bit for bit
    what the in-place division of the raw maps gives, with and without the in-schedule flip-TTA merge; the depth maps are untouched."""
    from benchkit.recipe import recipe_state_dict
    from benchkit.workload import make_cfg
    from model.smap import SMAP
    torch.manual_seed(0)
    cfg = make_cfg((16, 24))
    net = SMAP(cfg).eval()
    net.load_state_dict(recipe_state_dict(net.state_dict()))
    net = net.to(DEV)
    net.precision = precision
    from exps.stage3_root2.config import cfg as ds
    pair = (list(ds.DATASET.KEYPOINT.FLIP_ORDER) + [15 + c for c in ds.DATASET.PAF.FLIP_CHANNEL]) if flip else None
    x = torch.randn(3, 3, 64, 96, generator=torch.Generator().manual_seed(5)).to(DEV)
    raw = [t.clone() for t in net.engine(3, 64, 96, torch.device(DEV), flip_pair=pair).run(x)]
    eng = net.engine(3, 64, 96, torch.device(DEV), flip_pair=pair, scaled_hms=True)
    assert eng is not net.engine(3, 64, 96, torch.device(DEV), flip_pair=pair)
    assert sum(int(op.p.get("scale_hms", 0)) for op in eng.graph.ops) == 1
    got = [t.clone() for t in eng.run(x)]
    want = raw[0].cpu()            # on the HOST, as test_scale_hms_matches_reference_division: IEEE fp32 division (torch's GPU kernel multiplies
    want[:, :15] /= 255            # by the reciprocal of a scalar divisor, one ulp off in places); test.py:111-112
    want[:, 15:] /= 127
    assert raw[0].abs().max() > 0 and torch.equal(got[0].cpu(), want)
    import dapalib
    assert torch.equal(dapalib.scale_hms_(raw[0].clone()), got[0])        # = the stand-alone kernel the round-5 pipeline ran
    assert torch.equal(got[1], raw[1]) and torch.equal(got[2], raw[2])


def test_scale_hms_matches_reference_division():
    import dapalib
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 43, 128, 208, generator=g) * 100
    y = x.clone()
    y[:, :15] /= 255            # test.py:111-112
    y[:, 15:] /= 127
    z = dapalib.scale_hms_(x.to(DEV))
    assert torch.equal(z.cpu(), y)


def test_lift_and_refine_match_reference_golden(golden_dir):
    import dapalib
    from smap_amd.model.refinenet import RefineNet
    z = np.load(f"{golden_dir}/lift.npz")
    net = RefineNet().eval()
    net.load_state_dict(recipe_state_dict(net.state_dict()))
    wt, bs = net.folded(DEV)
    n = int(z["n_cases"])
    bodys = torch.zeros(n, 127, 15, 4)
    counts = torch.zeros(n, dtype=torch.int32)
    det, root, cams = [], [], []
    for c in range(n):
        p = f"c{c}_"
        b = z[p + "bodys"]
        bodys[c, :len(b)] = torch.from_numpy(b)
        counts[c] = len(b)
        det.append(expand(z[p + "det_c"], 0.05))
        root.append(expand(z[p + "root_c"], 0.002)[0])
        cams.append(z[p + "cam"])
    p2, p3, rz = dapalib.lift_batch(bodys.to(DEV), counts.to(DEV), torch.from_numpy(np.stack(det)).to(DEV),
                                    torch.from_numpy(np.stack(root)).to(DEV), np.stack(cams))
    ref = dapalib.refine_batch(p2, p3, counts.to(DEV), wt, bs)
    p2, p3, rz, ref = p2.cpu().numpy(), p3.cpu().numpy(), rz.cpu().numpy(), ref.cpu().numpy()
    for c in range(n):
        p = f"c{c}_"
        P = int(counts[c])
        # tolerance of the north star: 1e-3 m = 0.1 cm on 3D joints; the kernels are in fact bit-exact on lift
        assert np.abs(p3[c, :P] - z[p + "pred_3d"]).max() < 1e-4
        assert np.abs(rz[c, :P] - z[p + "root_z"]).max() < 1e-6
        assert np.abs(p2[c, :P] - z[p + "pred_2d"]).max() < 1e-4
        assert np.abs(ref[c, :P] - z[p + "refined"]).max() < 1e-2
        assert not p3[c, P:].any() and not ref[c, P:].any()
        o2, o3, orz = O.lift(z[p + "bodys"], det[c], root[c], cams[c])
        assert np.array_equal(p3[c, :P], o3) and np.array_equal(p2[c, :P], o2) and np.array_equal(rz[c, :P], orz)


def test_ground_truth_modes_match_reference_golden(golden_dir):
    """register_gt + f64 lifting + RefineNet on the device against the imported reference (lift_gt.npz) and,
    bit for bit, against the CPU oracle."""
    import dapalib
    from smap_amd.model.refinenet import RefineNet
    z = np.load(f"{golden_dir}/lift_gt.npz")
    net = RefineNet().eval()
    net.load_state_dict(recipe_state_dict(net.state_dict()))
    wt, bs = net.folded(DEV)
    n, G = int(z["n_cases"]), 20
    bodys = torch.zeros(n, 127, 15, 4)
    counts = torch.zeros(n, dtype=torch.int32)
    gt_roots = torch.zeros(n, G, 2)
    gt_counts = torch.zeros(n, dtype=torch.int32)
    det, root, cams = [], [], []
    for c in range(n):
        p = f"c{c}_"
        b = z[p + "bodys"]
        bodys[c, :len(b)] = torch.from_numpy(b)
        counts[c] = len(b)
        ann = z[p + "ann"]
        kept = ann[ann[:, 2, 3] > 1]                                   # test.py:76-80
        gt_roots[c, :len(kept)] = torch.from_numpy(kept[:, 2, :2])
        gt_counts[c] = len(kept)
        det.append(expand(z[p + "det_c"], 0.05))
        root.append(expand(z[p + "root_c"], 0.002)[0])
        cams.append(z[p + "cam"] if not int(z[p + "empty"]) else np.ones(9))
    matched, mc = dapalib.register_gt_batch(bodys.to(DEV), counts.to(DEV), gt_roots, gt_counts)
    p2, p3, rz = dapalib.lift_batch(matched, mc, torch.from_numpy(np.stack(det)).to(DEV),
                                    torch.from_numpy(np.stack(root)).to(DEV), np.stack(cams), gt_mode=True)
    ref = dapalib.refine_batch(p2, p3, mc, wt, bs)
    assert p2.dtype == torch.float64
    mc, p2, p3, rz, ref = mc.cpu().numpy(), p2.cpu().numpy(), p3.cpu().numpy(), rz.cpu().numpy(), ref.cpu().numpy()
    for c in range(n):
        p = f"c{c}_"
        if int(z[p + "empty"]):
            assert mc[c] == 0 and not p3[c].any()                      # the reference skips such frames
            continue
        P = len(z[p + "gt"])
        assert mc[c] == P
        assert np.array_equal(p2[c, :P], z[p + "matched"])             # matching + Z column, bit-exact
        assert np.array_equal(p3[c, :P], z[p + "pred_3d"]) and np.array_equal(rz[c, :P], z[p + "root_z"])
        assert np.abs(ref[c, :P] - z[p + "refined"]).max() < 1e-2
        assert not p3[c, P:].any() and not ref[c, P:].any()
        W, Bv = [w.t().contiguous().cpu().numpy() for w in wt], [b.cpu().numpy() for b in bs]
        assert np.array_equal(ref[c, :P], O.refine_gt(p2[c, :P], p3[c, :P], W, Bv))


def test_register_gt_random_vs_oracle():
    import dapalib
    rng = np.random.default_rng(5)
    B, G = 6, 24
    bodys = np.zeros((B, 127, 15, 4), np.float32)
    counts = np.array([0, 1, 9, 40, 127, 17], np.int32)
    gtc = np.array([3, 0, 24, 20, 11, 1], np.int32)
    gt = np.zeros((B, G, 2), np.float32)
    for b in range(B):
        bodys[b, :counts[b], :, :2] = rng.integers(0, 52, (counts[b], 15, 2)) * 4 + 0.5   # coarse grid: many exact ties
        bodys[b, :counts[b], :, 3] = rng.uniform(0.2, 1, (counts[b], 15))
        gt[b, :gtc[b]] = rng.integers(0, 104, (gtc[b], 2)) * 8 + 2.0
    m, mc = dapalib.register_gt_batch(torch.from_numpy(bodys).to(DEV), torch.from_numpy(counts).to(DEV), gt, gtc)
    m, mc = m.cpu().numpy(), mc.cpu().numpy()
    for b in range(B):
        if counts[b] == 0 or gtc[b] == 0:
            assert mc[b] == 0 and not m[b].any()
            continue
        want = O.register_gt(bodys[b, :counts[b]], gt[b, :gtc[b]])
        assert mc[b] == gtc[b] and np.array_equal(m[b, :gtc[b]], want) and not m[b, gtc[b]:].any()


def test_refinenet_forward_matches_golden(golden_dir):
    from model.refinenet import RefineNet
    z = np.load(f"{golden_dir}/refine.npz")
    net = RefineNet().eval()
    net.load_state_dict(recipe_state_dict(net.state_dict()))
    y = net.to(DEV)(torch.from_numpy(z["x"]).to(DEV)).cpu().numpy()
    assert np.abs(y - z["y"]).max() < 1e-4 * max(1.0, np.abs(z["y"]).max())


def test_flip_merge_kernel_matches_reference_loop():
    """smap_flip_merge == the reference's channel loop (test.py:55-70), bit for bit."""
    import dapalib
    from exps.stage3_root2.config import cfg
    from exps.stage3_root2.test_util import merge_flip
    g = torch.Generator().manual_seed(9)
    a = torch.randn(3, 43, 128, 208, generator=g) * 50
    b = torch.randn(3, 43, 128, 208, generator=g) * 50
    want = merge_flip(a.clone(), b, cfg)
    pair = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [15 + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
    got = dapalib.flip_merge_(a.to(DEV), b.to(DEV), pair)
    assert torch.equal(got.cpu(), want)
