"""End-to-end parity ON THE SAME IMAGES at BASELINE configs[2] (batch 8 x 3x512x832, the benchmarked schedule):

    path A  HIP backbone -> HIP association -> HIP lifting          (model.smap.SMAP + smap_amd.dapalib)
    path B  fp32 reference forward restated on the CPU -> oracle connect -> oracle lift
            (oracle/backbone_ref.py + oracle/smap_oracle.c = exps/stage3_root2/test.py:50-134)

north_star: "peak indices / limb assignments bit-exact, 3D joint coordinates within 1e-3 m" (= 0.1 cm in the reference's
cm units).  "Bit-exact peak indices" between two floating-point forwards is only defined above the forwards' own resolution
(benchkit/parity.py, "ties"): the assertion is that NO peak differs whose decision margin in the reference map exceeds 1e-6
of the map scale (peaks_clear_mismatch == 0; observed tie margins: 3e-8 .. 9e-8), that at most 3 candidates per 10 000 are such
near-ties (two per batch of ~2 800 candidates: 0 - 2 were observed per batch over 64 frames), and that every
3D joint of every paired skeleton is within 0.1 cm.  On the 16 frames of the seed-1234 batches the peak lists are in fact
identical; over 24 more frames one near-tie (margin ~1e-6) falls the other way.  Weights are the by-key recipe with calibrated heads (benchkit/workload.py::people_state_dict: ~24 peaks per
key-point channel, root depth ~3 m), in two flavours: "smooth" maps (coarse heads dominate) and "noise" maps (isolated
noise maxima: the fragile case for peak identity).

  precision "x3"  (fp16 hi/lo pairs, three MFMAs per K step)  MUST meet the north star -- asserted below;
  precision "f16" (fp16 storage, the fast mode)               is MEASURED and held to its own documented envelope
                                                              (it does not meet 0.1 cm at 3 m: ~0.3 cm mean).
Every run leaves its numbers in gpurun_out/e2e_parity_<precision>_<kind>.json (copied to profiles/ when committed).
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import make_cfg
from benchkit import parity
from benchkit.workload import people_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
B, H, W = 8, 512, 832
_REF = {}
KPT_GAIN_FLIP = float(os.environ.get("SMAP_TEST_KPT_GAIN_FLIP", "0.6"))   # see test_split_precision_flip_tta_end_to_end


def _setup(kind):
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = people_state_dict(net.state_dict(), kind)
    net.load_state_dict(sd)
    imgs = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1234))
    return net, sd, imgs


def _reference(kind, sd, imgs, cams):
    if kind not in _REF:                      # 8 fp32 CPU forwards: computed once per weight flavour
        _REF[kind] = parity.reference_path(sd, imgs, cams, threads=min(32, os.cpu_count() or 1))
    return _REF[kind]


def _dump(name, m):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        json.dump(m, f, indent=1)
    print(name, json.dumps(m))


def _assert_north_star(m):
    """Same peaks (above floating-point resolution), same skeletons and limbs, 3D joints within 1e-3 m."""
    assert m["peaks_clear_mismatch"] == 0, m                               # no peak differs that was not a floating-point tie
    # ... and such ties are rare: 3 per 10 000 of the pixels the NMS rule accepts (before the 127-cap, where the decisions are made)
    assert m["peaks_differing"] <= max(2, 3 * max(m["peak_candidates_ref"], m["peaks_ref"]) // 10000), m
    assert m["peaks_cap_shifted"] <= 2 * m["peaks_differing"], m           # the 127-cap moves only where a tie moved it
    assert m["peak_match"] >= 1.0 - 1e-3
    # a flipped near-tie can change the one skeleton it belongs to; everything that is paired must agree
    assert m["person_match"] >= 1.0 - 2.0 * max(m["peaks_differing"], 0) / max(m["persons_ref"], 1) - 1e-12
    assert m["limb_match"] >= 1.0 - 2.0 * max(m["peaks_differing"], 0) / max(m["persons_ref"], 1) - 1e-12
    if m["peaks_differing"] == 0:
        assert m["peak_match"] == 1.0 and m["person_match"] == 1.0 and m["limb_match"] == 1.0
    # north_star: 1e-3 m.  A joint beyond it must be a LIFTER TIE (benchkit/parity.py "lifter": a depth sample whose rounded
    # position sits within the centroid noise of an index step lands on the neighbouring pixel in one path) -- classified like the peak
    # ties, and as rare: at most 3 per 10 000 compared joints (or 2)
    assert m["joints_over_0.1cm_unexplained"] == 0, m
    assert m["lifter_tie_events"] <= max(2, 3 * m["joints_compared"] // 10000), m      # events = skeletons with a straddled step (each moves
    assert m["lifter_ties"] <= 5 * m["lifter_tie_events"]                                # the joints further down its limb chain: <= 5)
    # the tie bound is DERIVED (benchkit/parity.py::centroid_bounds: the centroid noise the measured map difference allows) and the
    # derivation is checked on every matched peak of the comparison; a tie joint's error stays under what its moved samples can cause
    assert m["centroid_peaks_checked"] >= 0.99 * m["peaks_ref"] and m["centroid_noise_max_over_bound"] <= 1.0, m
    assert m["lifter_ties_over_cap"] == 0, m
    # every joint beyond 1e-3 m is accounted for: a tie event moves <= 5 joints, a differing peak <= the 14 joints hanging off it
    assert m["joints_over_0.1cm"] <= 5 * m["lifter_tie_events"] + 14 * m["peaks_differing"], m
    assert m["joints_moved_after_peak_tie"] <= 14 * m["peaks_differing"], m
    # root depth: read at the SAME pixel in both paths unless the root coordinate straddles an integer (a root step = one more tie event)
    assert m["root_z_max_err_cm_same_pixel"] <= 0.1, m
    assert m["root_steps"] <= max(2, 3 * m["persons_ref"] // 10000), m
    if m["lifter_ties"] == 0 and m["peaks_differing"] == 0:
        assert m["max_joint_err_cm"] <= 0.1 and (m["root_steps"] > 0 or m["root_z_max_err_cm"] <= 0.1)
    assert max(m["map_rel_err_max"].values()) < 1e-4


@pytest.mark.parametrize("kind", ["smooth", "noise"])
def test_split_precision_meets_the_north_star_end_to_end(kind):
    from benchkit.workload import PEOPLE_CAM
    net, sd, imgs = _setup(kind)
    net.precision = "x3"
    net = net.to(DEV)
    cams = np.tile(np.asarray(PEOPLE_CAM, np.float64), (B, 1))
    hip = parity.hip_path(net, imgs.to(DEV), cams)
    ref = _reference(kind, sd, imgs, cams)
    m = parity.compare(hip, ref)
    m.update(precision="x3", weights=kind, batch=B)
    _dump(f"e2e_parity_x3_{kind}.json", m)
    assert m["persons_ref"] >= 8 * B and m["peaks_ref"] >= 100 * B, "the scene must contain people"
    _assert_north_star(m)
    if m["peaks_differing"] == 0:      # (the case on MI355X today) stronger than the 0.5 px pairing: lists equal entry by entry
        for a, b in zip(hip, ref):
            assert np.array_equal(a["peaks"][:, 0, 0], b["peaks"][:, 0, 0])
            assert np.abs(a["peaks"] - b["peaks"]).max() < 1e-3


@pytest.mark.parametrize("kind", ["smooth", "noise"])
def test_fp16_mode_measured_end_to_end_with_the_shipped_tile_table(kind):
    """The fast mode on the BENCHMARKED schedule: B = 8 hits smap_amd/tile_table.json (halo tiles 30..39 on their real
    shapes).  Its output is compared per frame with the fp32 reference; the end-to-end numbers are recorded."""
    from benchkit.workload import PEOPLE_CAM
    net, sd, imgs = _setup(kind)
    net.precision = "f16"
    net = net.to(DEV)
    eng = net.engine(B, H, W, torch.device(DEV))
    tiles = [op.p["tile"] for op in eng.graph.ops if op.kind == 0]
    assert any(t >= 30 for t in tiles), "B = 8 must run the measured tile table (halo-tiled 3x3 kernels)"
    cams = np.tile(np.asarray(PEOPLE_CAM, np.float64), (B, 1))
    hip = parity.hip_path(net, imgs.to(DEV), cams)
    ref = _reference(kind, sd, imgs, cams)
    m = parity.compare(hip, ref)
    m.update(precision="f16", weights=kind, batch=B, tiles=sorted(set(tiles)))
    _dump(f"e2e_parity_f16_{kind}.json", m)
    assert max(m["map_rel_err_max"].values()) < 1e-2           # fp16 storage tolerance, every frame of the batch
    assert m["peak_match"] > 0.9 and m["person_match"] > 0.9 and m["limb_match"] > 0.97
    assert m["mpjpe_cm"] < 1.0 and m["max_joint_err_cm"] < 3.0  # measured envelope (~0.3 / ~0.8 cm at Z ~ 3 m): NOT 0.1 cm


@pytest.mark.parametrize("seed", [11, 22, 33])
def test_split_precision_more_frames(seed):
    """24 more frames (other image seeds; the heads were calibrated on seed 1234's frame 0, so the peak counts vary more
    here): the same assertions -- how often does a 3e-6 perturbation of the maps flip a strict-> comparison?"""
    from benchkit.workload import PEOPLE_CAM
    net, sd, _ = _setup("smooth")
    imgs = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(seed))
    net.precision = "x3"
    net = net.to(DEV)
    cams = np.tile(np.asarray(PEOPLE_CAM, np.float64), (B, 1))
    hip = parity.hip_path(net, imgs.to(DEV), cams)
    ref = parity.reference_path(sd, imgs, cams, threads=min(32, os.cpu_count() or 1))
    m = parity.compare(hip, ref)
    m.update(precision="x3", weights="smooth", batch=B, image_seed=seed)
    _dump(f"e2e_parity_x3_smooth_seed{seed}.json", m)
    assert m["peaks_ref"] >= 50 * B
    _assert_north_star(m)


@pytest.mark.parametrize("gain", [KPT_GAIN_FLIP, 1.0])
@pytest.mark.parametrize("kind", ["smooth", "noise"])
def test_split_precision_flip_tta_end_to_end(kind, gain):
    """The reference's SHIPPED setting (test.sh: --do_flip 1) on the full batch of 8 frames, both weight flavours: HIP engine with
    the flip-TTA inside its schedule (16 frames of activations) -> association -> lifting vs the reference path with its second,
    mirrored forward and channel-loop merge (test.py:55-70) on the CPU.  gain = 1.0 is the workload as calibrated (the merge
    ADDS the key-point maps, so every channel sits at the 127-peak cap: the hardest case for peak identity); the same
    assertions hold there -- differing peaks are ties, joints beyond 1e-3 m are lifter ties."""
    from benchkit.workload import PEOPLE_CAM
    from exps.stage3_root2.config import cfg
    kpt = cfg.DATASET.KEYPOINT.NUM
    pair = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [kpt + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
    net, sd, imgs = _setup(kind)
    # the merge ADDS the key-point maps of the two passes (test.py:62-66: only the PAF channels are halved), which would put
    # every channel of this workload at the 127-peak cap; scaling the key-point heads by 0.6 (sum of two decorrelated passes:
    # mean x1.2, noise x0.85) brings the summed maps back near the calibrated level: candidates on both sides of the threshold
    for u in ("up2", "up3", "up4"):
        for t in ("weight", "bias"):
            k = f"stage2.upsample.{u}.res_conv2.bn.{t}"
            v = sd[k].clone()
            v[:kpt] *= gain
            sd[k] = v
    net.load_state_dict(sd)
    net.precision = "x3"
    net = net.to(DEV)
    cams = np.tile(np.asarray(PEOPLE_CAM, np.float64), (B, 1))
    hip = parity.hip_path(net, imgs.to(DEV), cams, flip_pair=pair)
    ref = parity.reference_path(sd, imgs, cams, threads=min(32, os.cpu_count() or 1), flip_pair=pair)     # 16 CPU forwards
    m = parity.compare(hip, ref)
    m.update(precision="x3", weights=kind, batch=B, flip_tta=True, kpt_gain=gain)
    _dump(f"e2e_parity_x3_flip_{kind}_gain{gain:g}.json", m)
    assert m["peaks_ref"] >= 20 * B and m["persons_ref"] >= B, m
    _assert_north_star(m)


def test_split_precision_with_refinenet_batch_of_8():
    """BASELINE configs[4]: batch 8 + RefineNet post-refinement, end to end against the reference path with the oracle's
    RefineNet (fp32 MLP; 1 ulp-level differences in the GEMM order): refined 3D joints within 1e-3 m."""
    from benchkit.workload import PEOPLE_CAM
    from benchkit.recipe import recipe_state_dict
    from model.refinenet import RefineNet
    net, sd, imgs = _setup("smooth")
    net.precision = "x3"
    net = net.to(DEV)
    torch.manual_seed(1)
    rnet = RefineNet().eval()
    rnet.load_state_dict(recipe_state_dict(rnet.state_dict()))
    wt, bs = rnet.folded("cpu")
    ref_w = ([w.t().contiguous().numpy() for w in wt], [b.numpy() for b in bs])
    cams = np.tile(np.asarray(PEOPLE_CAM, np.float64), (B, 1))
    hip = parity.hip_path(net, imgs.to(DEV), cams, refine=rnet.folded(DEV))
    ref = parity.reference_path(sd, imgs, cams, threads=min(32, os.cpu_count() or 1), refine=ref_w)
    m = parity.compare(hip, ref)
    m.update(precision="x3", weights="smooth", batch=B, refinenet=True)
    _dump("e2e_parity_x3_refinenet.json", m)
    assert m["persons_ref"] >= 8 * B
    _assert_north_star(m)


def test_fp16_flip_batch_of_16_agrees_with_batch_of_8():
    """2B = 16 is the flip-TTA batch of the B = 8 pipeline: another tiling of M, same numbers within fp16 tolerance."""
    net, sd, imgs = _setup("smooth")
    net.precision = "f16"
    net = net.to(DEV)
    x = imgs.to(DEV)
    a = [t.clone() for t in net(x)]
    b = net(torch.cat([x, torch.flip(x, [-1])], 0))
    for u, v in zip(a, b):
        assert (u - v[:B]).abs().max().item() <= 1e-2 * u.abs().max().item()
