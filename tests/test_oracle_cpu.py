"""CPU suite: pins the oracle (oracle/) against the golden vectors produced by importing the
reference's Python (tests/golden/gen_golden.py), and against known answers that follow from
the reference's kernel semantics (strict >, global scan rank, int(a+.5), caps)."""
import numpy as np
import pytest
import torch

from helpers import make_cfg, synth_scene, noise_scene, PAIRS
from fixture_maps import expand
from recipe import recipe_state_dict
from oracle import oracle_lib as O


# ------------------------------------------------------------------ lift / refine (golden)
def _refine_folded():
    from smap_amd.model.refinenet import RefineNet
    net = RefineNet().eval()
    net.load_state_dict(recipe_state_dict(net.state_dict()))
    wt, bs = net.folded("cpu")
    return [w.t().contiguous().numpy() for w in wt], [b.numpy() for b in bs], net


def test_lift_matches_reference_golden(golden_dir):
    z = np.load(f"{golden_dir}/lift.npz")
    for c in range(int(z["n_cases"])):
        p = f"c{c}_"
        det, root = expand(z[p + "det_c"], 0.05), expand(z[p + "root_c"], 0.002)[0]
        p2, p3, rz = O.lift(z[p + "bodys"], det, root, z[p + "cam"])
        # bit-exact against the reference's numpy post-process
        assert np.array_equal(p2, z[p + "pred_2d"])
        assert np.array_equal(p3, z[p + "pred_3d"])
        assert np.array_equal(rz, z[p + "root_z"])


def test_refine_matches_reference_golden(golden_dir):
    z = np.load(f"{golden_dir}/lift.npz")
    W, B, _ = _refine_folded()
    for c in range(int(z["n_cases"])):
        p = f"c{c}_"
        out = O.refine(z[p + "pred_2d"], z[p + "pred_3d"], W, B)
        # 3D joints within 1e-3 m == 0.1 cm (north_star); observed ~1 fp32 ulp
        assert np.abs(out - z[p + "refined"]).max() < 1e-2


# ------------------------------------------------------------------ ground-truth modes (golden)
def test_register_gt_lift_refine_match_reference_golden(golden_dir):
    """generate_result / generate_train: register_pred with ground truth, f64 lifting, RefineNet -- bit-exact
    (matching, 2D, 3D, root depth) against the imported reference (tests/golden/gen_golden_gt.py)."""
    z = np.load(f"{golden_dir}/lift_gt.npz")
    W, B, _ = _refine_folded()
    seen_unmatched = seen_matched = 0
    for c in range(int(z["n_cases"])):
        p = f"c{c}_"
        if int(z[p + "empty"]):
            continue
        gt = z[p + "gt"]
        m = O.register_gt(z[p + "bodys"], gt[:, 2, :2])
        det, root = expand(z[p + "det_c"], 0.05), expand(z[p + "root_c"], 0.002)[0]
        p2, p3, rz = O.lift_gt(m, det, root, z[p + "cam"])
        assert p2.dtype == np.float64 and np.array_equal(p2, z[p + "matched"])
        assert np.array_equal(p3, z[p + "pred_3d"]) and np.array_equal(rz, z[p + "root_z"])
        out = O.refine_gt(p2, p3, W, B)
        assert np.abs(out - z[p + "refined"]).max() < 1e-2
        assert np.array_equal(out[:, :, 3], z[p + "refined"][:, :, 3])        # score column: 0 for unmatched persons
        seen_unmatched += int((p2[:, 2, 3] == 0).sum())
        seen_matched += int((p2[:, 2, 3] != 0).sum())
    assert seen_unmatched > 0 and seen_matched > 0


def test_register_gt_known_answers():
    bodys = np.zeros((3, 15, 4), np.float32)
    bodys[:, 2, 3] = 1.0
    bodys[0, 2, :2] = (10, 10)          # x4 -> (40, 40)
    bodys[1, 2, :2] = (12, 10)          # x4 -> (48, 40)
    bodys[2, 2, :2] = (100, 60)         # x4 -> (400, 240)
    bodys[:, 0, 0] = (1, 2, 3)          # marks which prediction landed where
    # annotation 0 and 1 sit exactly between predictions 0 and 1 (distance 4 each): ties go in row-major order, so
    # annotation 0 takes prediction 0 and annotation 1 then gets prediction 1; annotation 2 is 29.x px from
    # prediction 2 (accepted), annotation 3 is exactly 30 px away from it (strict <: rejected, and it is taken)
    gt = np.array([[44, 40], [44, 40], [400, 269.5], [430, 240]], np.float32)
    m = O.register_gt(bodys, gt)
    assert m[:, 0, 0].tolist() == [1.0, 2.0, 3.0, 0.0]
    assert not m[3].any()
    # no prediction at all -> zeros (the caller skips the frame)
    assert not O.register_gt(np.zeros((0, 15, 4), np.float32), gt).any()


def test_convert_matches_reference_golden(golden_dir, tmp_path):
    """lib/eval/convert.py against the imported reference's .mat output (tests/golden/gen_golden_gt.py)."""
    import json
    import scipy.io as scio
    from lib.eval.convert import convert
    z = np.load(f"{golden_dir}/convert.npz")
    src = tmp_path / "result.json"
    src.write_text(bytes(z["json"]).decode())
    p3, p2 = convert(str(src), out_dir=str(tmp_path))
    m3 = scio.loadmat(str(tmp_path / "pose3d.mat"))["preds_3d_kpt"]
    m2 = scio.loadmat(str(tmp_path / "pose2d.mat"))["preds_2d_kpt"]
    assert list(m3.dtype.names) == [str(n) for n in z["names"]]
    for k, name in enumerate(z["names"]):
        assert np.array_equal(p3[str(name)], z[f"p3_{k}"]) and np.array_equal(p2[str(name)], z[f"p2_{k}"])
        assert np.array_equal(m3[str(name)][0, 0], z[f"p3_{k}"]) and np.array_equal(m2[str(name)][0, 0], z[f"p2_{k}"])
    # the writer's own key names are accepted too
    doc = json.loads(bytes(z["json"]).decode())
    for r in doc["3d_pairs"]:
        del r["pred"], r["gt"]
    src.write_text(json.dumps(doc))
    q3, _ = convert(str(src), out_dir=str(tmp_path))
    assert all(np.array_equal(q3[k], p3[k]) for k in p3)
    doc["3d_pairs"][0]["image_path"] = "/data/TS21/img.jpg"
    src.write_text(json.dumps(doc))
    with pytest.raises(NotImplementedError):
        convert(str(src), out_dir=str(tmp_path))


def test_refinenet_module_keys(golden_dir):
    z = np.load(f"{golden_dir}/refine.npz")
    _, _, net = _refine_folded()
    assert len(net.state_dict()) == int(z["n_keys"]) == 30


# ------------------------------------------------------------------ backbone (golden)
@pytest.fixture(scope="module")
def small_model():
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    sd = recipe_state_dict(net.state_dict())
    net.load_state_dict(sd)
    return net, sd


def test_module_keys_and_default_init_match_reference(golden_dir):
    from smap_amd.model.smap import SMAP
    d = np.load(f"{golden_dir}/default_init.npz")
    torch.manual_seed(0)
    sd = SMAP(make_cfg((128, 208))).state_dict()
    assert list(sd.keys()) == [str(k) for k in d["all_keys"]]          # 1876 keys, same order
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in d["all_shapes"]]
    for k, s, a in zip(d["keys"], d["sums"], d["abssums"]):            # same RNG consumption order
        assert float(sd[str(k)].double().sum()) == pytest.approx(float(s), abs=1e-9)
        assert float(sd[str(k)].double().abs().sum()) == pytest.approx(float(a), abs=1e-9)
    assert sum(p.numel() for p in SMAP(make_cfg()).parameters()) == 91318696


def test_backbone_ref_matches_reference_golden(golden_dir, small_model):
    from oracle.backbone_ref import smap_forward
    z = np.load(f"{golden_dir}/backbone_small.npz")
    _, sd = small_model
    with torch.no_grad():
        h, d, r = smap_forward(sd, torch.from_numpy(z["x"]))
    for a, k in ((h, "hms"), (d, "det_d"), (r, "root_d")):
        assert np.abs(a.numpy() - z[k]).max() <= 1e-5 * np.abs(z[k]).max()


def check_against_full_size_digest(z, outs, tol):
    """outs = (hms, det_d, root_d) [1,C,128,208] fp32 tensors vs tests/golden/backbone_full.npz (digest of the IMPORTED
    reference model's outputs at 1x3x512x832, tests/golden/gen_golden_full.py): 1024 sampled positions per output within
    tol of the output's max |.|, per-channel sums within tol of the channel's sum of |.|.  Shared with the GPU test."""
    worst = {}
    for name, t in zip(("hms", "det_d", "root_d"), outs):
        a = t[0].detach().cpu().numpy()
        assert tuple(a.shape) == tuple(z[name + "_shape"]), name
        err = np.abs(a.reshape(-1)[z[name + "_idx"]] - z[name + "_val"]).max() / float(z[name + "_absmax"])
        chs = np.abs(a.astype(np.float64).sum((1, 2)) - z[name + "_chsum"]) / z[name + "_chabs"]
        worst[name] = (float(err), float(chs.max()))
        assert err <= tol, (name, "sampled positions", err)
        assert chs.max() <= tol, (name, "channel sums", chs.max())
    return worst


def full_size_input(z):
    x = torch.randn(1, 3, 512, 832, generator=torch.Generator().manual_seed(1234))
    assert abs(float(x.double().sum()) - float(z["x_sum"])) < 1e-6 and abs(float(x.double().abs().sum()) - float(z["x_abssum"])) < 1e-3
    return x


def test_backbone_ref_matches_reference_at_full_size(golden_dir):
    """the backbone oracle pinned to the reference at the BENCHMARKED size, not only at 64x96: the imported
    model.smap.SMAP at 1x3x512x832 (digest: per-channel sums + 1024 sampled positions per output) vs oracle/backbone_ref.py
    on the same recipe weights and input.  Not bit-exact (oneDNN's summation order depends on the thread count): 1e-5."""
    from smap_amd.model.smap import SMAP
    from oracle.backbone_ref import smap_forward
    z = np.load(f"{golden_dir}/backbone_full.npz")
    torch.manual_seed(0)
    sd = recipe_state_dict(SMAP(make_cfg((128, 208))).state_dict())
    x = full_size_input(z)
    with torch.no_grad():
        outs = smap_forward(sd, x)
    worst = check_against_full_size_digest(z, outs, 1e-5)
    print("full-size oracle vs imported reference (sampled, channel sums):", worst)


def test_default_schedule_wiring_matches_reference_golden(golden_dir, small_model):
    """The DEFAULT schedule of round 6 -- shortcut convs inside their blocks' last 1x1 (conv_cat), skip1 + skip2 as one launch and one tensor
    (conv_relusum), the root-depth head as tap dots + stencil (conv_tapdot / tapsum) -- interpreted on CPU, against the golden outputs of the
    imported reference model."""
    from smap_amd.engine import Graph, OP_TAPSUM
    from oracle.graph_interp import run_graph
    z = np.load(f"{golden_dir}/backbone_small.npz")
    _, sd = small_model
    g = Graph(sd, 2, 64, 96, keep_ref=True)
    g.allocate()
    assert sum(1 for op in g.ops if "cat" in op.p and op.p["cat"].get("relusum")) == 8 and sum(1 for op in g.ops if op.kind == OP_TAPSUM) == 1
    assert sum(1 for op in g.ops if "cat" in op.p and not op.p["cat"].get("relusum")) == 12 and sum(1 for op in g.ops if "tap" in op.p) == 1
    with torch.no_grad():
        outs = run_graph(g, torch.from_numpy(z["x"]), quantize=False)
        outs_q = run_graph(g, torch.from_numpy(z["x"]), quantize=True)
    for a, q, k in zip(outs, outs_q, ("hms", "det_d", "root_d")):
        ref = z[k]
        assert np.abs(a.numpy() - ref).max() <= 2e-5 * np.abs(ref).max(), k
        assert np.abs(q.numpy() - ref).max() <= 2e-2 * np.abs(ref).max(), k


@pytest.mark.parametrize("merge", ["1", "2", "0"])
def test_schedule_wiring_matches_reference_golden(golden_dir, small_model, monkeypatch, merge):
    """The engine's op list, interpreted in fp32 on CPU, reproduces the reference outputs:
    folding, dead-head removal, commuted up_conv, merged heads, epilogue skip adds -- with the shared-input 1x1s of every
    Upsample_unit (smap.py:210-241) as one launch with several outputs (default) and as one launch each."""
    from smap_amd.engine import Graph
    from oracle.graph_interp import run_graph
    z = np.load(f"{golden_dir}/backbone_small.npz")
    _, sd = small_model
    monkeypatch.setenv("SMAP_MERGE_1X1", merge)
    monkeypatch.setenv("SMAP_SKIPSUM", "0")              # (the default -- skip1 + skip2 as one launch -- has its own test below)
    g = Graph(sd, 2, 64, 96, keep_ref=True)
    g.allocate()
    # 203 convs + stem + maxpool + 3 head sums; "2" (every shared-input pair): 2 launches fewer per Upsample_unit of stages 0 / 1, 1 fewer
    # for up2 / up3 of stage 2; "1" (default: the merges that measured faster) keeps u_skip | skip1 apart where u_skip has the bilinear add
    saved = {"2": 18, "1": 12, "0": 0}[merge]
    # ... and (round 6) the 12 shortcut convs -- first block of every layer, 3 stages -- inside their blocks' last 1x1 (Graph.conv_cat): 203 -> 191 convs
    assert len(g.ops) == 208 - 12 - saved and sum(len(op.outs) for op in g.ops) == saved and sum(1 for op in g.ops if "cat" in op.p) == 12
    with torch.no_grad():
        outs = run_graph(g, torch.from_numpy(z["x"]), quantize=False)
        outs_q = run_graph(g, torch.from_numpy(z["x"]), quantize=True)
    for a, q, k in zip(outs, outs_q, ("hms", "det_d", "root_d")):
        ref = z[k]
        assert np.abs(a.numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
        assert np.abs(q.numpy() - ref).max() <= 5e-3 * np.abs(ref).max()   # fp16 storage budget


def test_arena_allocation_has_no_live_overlap(small_model):
    from smap_amd.engine import Graph
    _, sd = small_model
    g = Graph(sd, 1, 64, 96)
    g.allocate(reuse=True)
    ts = [t for t in g.tensors if t.off >= 0]
    assert len(ts) == len(g.tensors)
    for i, a in enumerate(ts):
        for b in ts[i + 1:]:
            live = not (a.last < b.first or b.last < a.first)
            mem = not (a.off + a.nbytes <= b.off or b.off + b.nbytes <= a.off)
            assert not (live and mem), (a.name, b.name)
    g2 = Graph(sd, 1, 64, 96)
    assert g2.allocate(reuse=False) >= g.arena_bytes


# ------------------------------------------------------------------ association known answers
def _blank(H=128, W=208):
    return np.zeros((43, H, W), np.float32)


def test_nms_single_symmetric_blob():
    hms = _blank()
    yy, xx = np.mgrid[0:128, 0:208]
    hms[3] = np.exp(-((xx - 50) ** 2 + (yy - 40) ** 2) / (2 * 1.5 ** 2)).astype(np.float32)
    pk = O.nms(hms)
    assert pk[3, 0, 0] == 1 and all(pk[c, 0, 0] == 0 for c in range(15) if c != 3)
    # symmetric blob: centroid == centre, +0.5 offset (nmsBase.cu:127-128)
    assert abs(pk[3, 1, 0] - 50.5) < 1e-4 and abs(pk[3, 1, 1] - 40.5) < 1e-4
    assert pk[3, 1, 2] == np.float32(1.0)


def test_nms_plateau_and_border_and_threshold():
    hms = _blank()
    hms[0, 10, 10] = hms[0, 10, 11] = 0.9          # two equal neighbours: strict > -> no peak
    hms[1, 0, 5] = 0.9                              # border row never registers
    hms[2, 20, 20] = 0.2                            # == threshold: not > 0.2
    hms[4, 20, 20] = np.float32(0.2) + np.float32(1e-6)
    pk = O.nms(hms)
    assert pk[0, 0, 0] == 0 and pk[1, 0, 0] == 0 and pk[2, 0, 0] == 0 and pk[4, 0, 0] == 1


def test_nms_raster_order_and_cap():
    hms = _blank()
    pts = [(y, x) for y in range(4, 124, 8) for x in range(4, 204, 8)][:300]   # >= 8 px apart: 7x7 windows isolated
    for i, (y, x) in enumerate(pts):
        hms[7, y, x] = 0.3 + 0.001 * (i % 50)
    pk = O.nms(hms)
    assert pk[7, 0, 0] == 127                       # truncated to maxPeaks (nmsBase.cu:131-133)
    for r in range(127):                            # r-th peak in raster (y-major) order
        y, x = pts[r]
        assert pk[7, r + 1, 2] == hms[7, y, x]
        assert abs(pk[7, r + 1, 0] - (x + 0.5)) < 1e-5 and abs(pk[7, r + 1, 1] - (y + 0.5)) < 1e-5


def test_paf_score_straight_limb_and_fallbacks():
    hms = _blank()
    hms[0, 30, 40] = 1.0      # neck
    hms[1, 30, 80] = 1.0      # head, 40 px to the right
    hms[15, 28:33, 38:84] = 1.0      # limb 0 PAF-x ribbon, unit vector (1,0)
    hms[2, 60, 100] = 1.0     # pelvis far away, no PAF evidence on limb 1
    pk = O.nms(hms)
    sc = O.paf_score(hms, pk)
    assert sc.shape == (14, 127, 127)
    assert sc[0, 0, 0] == pytest.approx(1.0, abs=1e-6)          # every sample agrees with the limb
    assert sc[0, 0, 1] == -1 and sc[0, 1, 0] == -1              # no such peak pair
    # limb 1 (neck->pelvis): no PAF evidence and far apart -> -1
    assert sc[1, 0, 0] == -1


def test_paf_score_close_points_get_default():
    hms = _blank()
    hms[0, 30, 40] = 1.0
    hms[0, 30, 41] = 0.5      # makes the neck centroid sub-pixel, still one peak
    hms[1, 31, 40] = 1.0      # head 1 px away, PAF empty -> distance fallback (bodyPartConnectorBase.cu:56-59)
    sc = O.paf_score(hms, O.nms(hms))
    assert sc[0, 0, 0] == np.float32(np.float32(0.1) + 1e-6)


def test_group_depth_order_and_empty():
    hms, rdepth, joints, depths = synth_scene(3, seed=5, noise=0.0, drop=0.0)
    bodys, pk, sc = O.connect(hms, rdepth)
    assert bodys.shape == (3, 15, 4)
    assert np.all(bodys[:, :, 2] == 0)                             # column 2 untouched (association.cpp:151)
    order = np.argsort(depths, kind="stable")
    for i, p in enumerate(order):                                  # near -> far, all joints found
        assert np.abs(bodys[i, :, :2] - (joints[p] + 0.5)).max() < 1.5
        assert np.all(bodys[i, :, 3] > 0.2)
    empty, _, _ = O.connect(_blank(), np.ones((128, 208), np.float32))
    assert empty.shape == (0, 15, 4)


def test_group_each_candidate_used_once():
    hms, rdepth = noise_scene(3)
    bodys, pk, sc = O.connect(hms, rdepth)
    assert bodys.shape[0] == 127
    for j in range(15):
        got = bodys[bodys[:, j, 3] > 0][:, j, :2]
        assert len(np.unique(got, axis=0)) == len(got)


def test_depth_sort_matches_torch_sort():
    """association.cpp:144 sorts root depths with Tensor::sort(0, false): torch's CPU kernel is
    std::sort over (value, index) pairs, NOT stable.  The oracle restates libstdc++'s introsort;
    pin it against torch itself on tie-heavy, ordered, NaN and adversarial inputs."""
    rng = np.random.default_rng(0)

    def check(d):
        d = np.asarray(d, np.float32)
        v, i = torch.from_numpy(d.copy()).sort(0, False)
        oi, ov = O.sort_depth(d)
        assert np.array_equal(i.numpy(), oi), (len(d), d[:8])
        assert np.array_equal(v.numpy().view(np.uint32), ov.view(np.uint32))

    for n in range(0, 128):
        for _ in range(12):
            k = rng.integers(1, max(2, n))
            vals = rng.uniform(0.1, 2, k).astype(np.float32)
            check(vals[rng.integers(0, k, n)] if n else np.zeros(0, np.float32))
        check(np.arange(n))
        check(np.arange(n)[::-1].copy())
        check(np.ones(n))
        check(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]))
        x = rng.uniform(0, 1, n).astype(np.float32)
        if n > 3:
            x[rng.integers(0, n, 2)] = np.nan
        check(x)


def test_group_tie_order_follows_torch_sort():
    hms, rdepth, _, _ = synth_scene(20, seed=320)      # overlapping depth discs -> equal root depths
    pk = O.nms(hms)
    P = int(pk[2, 0, 0])
    d = np.array([rdepth[int(pk[2, i + 1, 1]), int(pk[2, i + 1, 0])] for i in range(P)], np.float32)
    assert len(np.unique(d)) < P                       # the scene really has ties
    _, idx = torch.from_numpy(d).sort(0, False)
    bodys, _, _ = O.connect(hms, rdepth)
    assert np.array_equal(bodys[:, 2, :2], pk[2, 1 + idx.numpy(), :2])
