"""CPU tests of the host logic added in round 2: split-precision weight packing and schedule emission, the in-schedule
flip-TTA op list, tile tables, lazy result records, the end-to-end parity checker on synthetic inputs."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import make_cfg
from recipe import recipe_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def small_sd():
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    return recipe_state_dict(SMAP(make_cfg((16, 24))).state_dict())


def test_split_f16_reconstructs_to_22_bits():
    from smap_amd.engine import split_f16
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 576, generator=g, dtype=torch.float64) * 0.03
    w[0, 0], w[0, 1] = 0.0, 1e-7                                   # vanishing weights survive too
    hi, lo, inv = split_f16(w)
    assert hi.dtype == lo.dtype == torch.float16 and torch.isfinite(hi).all() and torch.isfinite(lo).all()
    s = 1.0 / inv
    assert 2 ** 13 <= w.abs().max().item() * s < 2 ** 14 and float(np.log2(s)).is_integer()
    back = (hi.double() + lo.double()) * inv
    assert ((back - w).abs() <= w.abs() * 2.0 ** -21 + 2.0 ** -24 * inv).all()
    # plain fp16 is three orders of magnitude coarser on the same data
    assert (w.half().double() - w).abs().max() > 100 * (back - w).abs().max()
    z = split_f16(torch.zeros(4, 8, dtype=torch.float64))
    assert z[2] == 1.0 and not z[0].any() and not z[1].any()


@pytest.mark.parametrize("blocks", [False, True], ids=["layer_by_layer", "whole_blocks"])
def test_x3_graph_emits_split_strides_and_weights(small_sd, monkeypatch, blocks):
    from smap_amd.engine import Graph, OP_CONV, OP_HEADSUM, OP_MAXPOOL, OP_STEM, X3_TILES, ZERO_PAGE
    if not blocks:                      # the layer-by-layer schedule: op for op the fp16 one (the default fuses layer1's Bottlenecks)
        monkeypatch.setenv("SMAP_BLOCK", "")
        monkeypatch.setenv("SMAP_BLOCK_FIRST", "")
    else:                               # (named explicitly: a schedule this small drops layer2's whole-block launches by default)
        monkeypatch.setenv("SMAP_BLOCK", "64:91,128:94")
    monkeypatch.setenv("SMAP_SPLITK", "0")          # (this test is about the split-precision STORAGE: no split-K scratch, no lanes, whose
    monkeypatch.setenv("SMAP_LANES", "0")           #  extended lifetimes would blur the arena comparison at the end)
    g16 = Graph(small_sd, 2, 64, 96)
    g = Graph(small_sd, 2, 64, 96, precision="x3")
    g.allocate()
    ops = g.emit()
    n_blk = sum(1 for op in g.ops if "head" in op.p)
    assert n_blk == (18 if blocks else 0)                 # 3 stages x (first block + two identity blocks of layer1 + three identity blocks of layer2)
    # (a first block is 3 launches layer by layer since round 6 -- c1, c2, c3 with the shortcut conv inside, Graph.conv_cat -- and 1 as a whole block)
    assert len(g.ops) == len(g16.ops) - (3 * 2 + 2 * 6 + 2 * 9 if blocks else 0) and g.flops == g16.flops
    ncat = lambda gr: sum(1 for op in gr.ops if "cat" in op.p and not op.p["cat"].get("relusum"))
    assert ncat(g) == (9 if blocks else 12) and ncat(g16) == 12
    # ... and the eight skip pairs of stages 0 / 1 (skip1 on the unit's input + skip2 on its output) are one launch, one tensor each (Graph.conv_relusum)
    assert sum(1 for op in g.ops if "cat" in op.p and op.p["cat"].get("relusum")) == 8 and not any(t.name.endswith((".skip1", ".skip2")) for t in g.tensors)
    assert not any(op.out is not None and op.out.name.endswith(".downsample") for op in g.ops)
    assert g.weight_blob().numel() > 1.9 * g16.weight_blob().numel()
    for op, o in zip(g.ops, ops):
        assert o.precision == 1
        if op.kind == OP_CONV:
            x, y = op.inp, op.out
            assert x.planes == 2 and o.in_stride_c == 2 * x.C and o.in_c_off + o.Cin <= x.C
            assert o.tile in X3_TILES + (3,) + tuple(range(30, 46)) + (90, 91, 92, 93, 94) and o.acc_scale > 0
            assert (y.planes, o.out_stride_c) == ((1, y.C) if o.out_fp32 else (2, 2 * y.C))
            assert o.in_off >= ZERO_PAGE and o.Cin * 2 + o.in_stride_c + 16 <= ZERO_PAGE
            for t in (op.res, op.add1, op.add2):
                assert t is None or t.planes == 2
        elif op.kind == OP_STEM:
            assert o.acc_scale > 0 and op.out.planes == 2
        elif op.kind == OP_MAXPOOL:
            assert op.inp.planes == op.out.planes == 2
        elif op.kind == OP_HEADSUM:
            assert all(t.esize == 4 and t.planes == 1 for t in op.aux)
    # arena: split tensors take twice the bytes (fewer of them are alive at once when layer1's intermediates stay on chip)
    g16.allocate()
    assert g.arena_bytes > (1.6 if blocks else 1.8) * g16.arena_bytes


def test_flip_graph_runs_two_B_frames_and_merges_in_the_head_sum(small_sd):
    from smap_amd.engine import Graph, OP_CONV, OP_HEADSUM, OP_STEM, OP_TAPSUM
    pair = list(range(43))[::-1]
    g = Graph(small_sd, 3, 64, 96, flip_pair=pair)
    g.allocate()
    ops = g.emit()
    assert g.B == 6 and g.frames == 3 and g.out_bytes == 3 * 58 * 16 * 24 * 4
    kinds = [op.kind for op in g.ops]
    stem = ops[kinds.index(OP_STEM)]
    assert stem.B == 6 and stem.flip_from == 3
    heads = [o for o in ops if o.kind in (OP_HEADSUM, OP_TAPSUM)]      # hms, det_d: head sums; root_d: the stencil half of its 3x3 (Graph.tapsum)
    assert [h.kind for h in heads] == [OP_HEADSUM, OP_HEADSUM, OP_TAPSUM]
    assert [h.B for h in heads] == [3, 3, 3] and [h.flip_from for h in heads] == [3, 0, 0]
    assert heads[0].in_c_off == 15 and heads[0].w_off >= 0
    blob = g.weight_blob()
    tab = blob[heads[0].w_off:heads[0].w_off + 43 * 4].view(torch.int32).tolist()
    assert tab == pair
    convs = {op.out.name: o for op, o in zip(g.ops, ops) if op.kind == OP_CONV}
    depth = [n for n in convs if n.endswith(".res_d") or n.endswith(".res_rd_t")]
    assert len(depth) == 2 and all(convs[n].B == 3 for n in depth)          # mirrored half of the depth heads: never read
    assert all(o.B == 6 for n, o in convs.items() if n not in depth)
    with pytest.raises(AssertionError):
        Graph(small_sd, 3, 64, 96, flip_pair=[0, 1, 2])


def test_shipped_launcher_settings_are_plannable():
    """exps/stage3_root2/test.sh runs --batch_size 16 --do_flip 1 in the default split precision: 32 frames of activations, a
    5.6 GiB arena.  Until round 3 the conv kernels addressed the whole arena with 32-bit offsets (4 GiB) and the batch ran as
    2 x 8; now a launch addresses its input from the input's 4 GiB WINDOW, the arena reserves a zero page at the start of every
    window, and the shipped setting plans as ONE schedule.  The limit that remains is one tensor per window (64 frames of
    the 768-channel head tensor do not fit), and the graph says so in words."""
    from types import SimpleNamespace as NS
    from smap_amd.engine import ArenaTooLarge, Graph, WINDOW, ZERO_PAGE, OP_CONV
    from smap_amd.model.smap import SMAP
    sh = open(os.path.join(ROOT, "exps", "stage3_root2", "test.sh")).read()
    assert "--batch_size 16" in sh and "--do_flip 1" in sh
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
             OUTPUT_SHAPE=(128, 208), LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    sd = SMAP(cfg).state_dict()
    g = Graph(sd, 16, 512, 832, precision="x3", flip_pair=list(range(43)))
    assert g.allocate() > (1 << 32)                                         # beyond what 32-bit arena offsets could address
    windows = set()
    for t in g.tensors:                                                     # no tensor on a zero page, none across a window boundary
        assert t.off % WINDOW >= ZERO_PAGE and t.off // WINDOW == (t.off + t.nbytes - 1) // WINDOW, t.name
    for op, o in zip(g.ops, g.emit()):
        if op.kind == OP_CONV:
            windows.add(o.in_off // WINDOW)
            assert o.in_off % WINDOW + op.inp.nbytes <= (1 << 32)
    assert len(windows) >= 2, "the test must exercise a launch whose input lies beyond the first window"
    live = sorted((t.first, t.last, t.off, t.off + t.nbytes) for t in g.tensors)
    for i, a in enumerate(live):                                            # live ranges never share bytes
        for b in live[i + 1:]:
            if b[0] > a[1]:
                break
            assert a[3] <= b[2] or b[3] <= a[2] or a[1] < b[0]
    # the LIBRARY agrees with the allocator's idea of windows and zero pages (smap_plan_create runs without a GPU): a library
    # built with another window size than engine.WINDOW refuses these ops -- round 4 lost a GPU visit to exactly that
    import ctypes as C
    from smap_amd import lib as L
    h = C.c_void_p()
    assert L.load().smap_plan_create(g.emit(), len(g.ops), C.byref(h)) == 0
    L.load().smap_plan_destroy(h)
    # (the widest tensor is the 512-channel head tensor since round 6 -- the root-depth head no longer stores its 256 channels --: 78 frames per window)
    Graph(sd, 32, 512, 832, precision="x3", flip_pair=list(range(43))).allocate()
    with pytest.raises(ArenaTooLarge, match="smaller batch"):
        Graph(sd, 40, 512, 832, precision="x3", flip_pair=list(range(43))).allocate()
    with pytest.raises(ValueError):
        Graph(sd, 1, 64, 96, flip_pair=[0] * 43)             # not a permutation


def test_tile_tables_name_existing_tiles():
    from smap_amd.engine import TILES, X3_TILES, _table_entry, tile_family
    t16 = json.load(open(os.path.join(ROOT, "smap_amd", "tile_table.json")))
    tx3 = json.load(open(os.path.join(ROOT, "smap_amd", "tile_table_x3.json")))
    assert t16 and all(t in TILES for v in t16.values() for t in _table_entry(v))
    assert tx3 and all(t in X3_TILES + (3,) + tuple(range(30, 46)) for v in tx3.values() for t in _table_entry(v))
    for key, v in tx3.items():
        f = key.split(",")                                                   # optional 8th field: "up" (ops with a fused bilinear add)
        merged = "+" in f[4]                                                 # "c0+c1[+c2]": a merged 1x1 launch (Graph.conv_seg, tools/autotune_seg.py)
        B, H, W, cin, k, s = map(int, f[:4] + f[5:7])
        cout = sum(map(int, f[4].split("+")))
        assert f[7:] in ([], ["up"])
        assert not merged or (k == 1 and s == 1 and all(tile_family(t) == "igemm" for t in _table_entry(v)))
        for t in _table_entry(v):
            assert B in (1, 8, 16) and (t < 30 or t >= 40 or (k == 3 and s == 1))   # (1 = configs[1], 16 = the flip-TTA schedule of batch 8); halo tiles: plain 3x3 stride 1 only
            assert cout > 32 or t in (3, 38, 39)
        assert not 60 <= _table_entry(v)[-1] < 80                            # a ranked list ends on a tile that takes every op


def test_tile_geometry_tables_agree_with_the_library():
    """engine.py packs the weight blob per tile (BN rows x BK halves blocks); the kernels' template arguments are the truth."""
    import ctypes as C
    from smap_amd import lib as L
    from smap_amd.engine import TILES, tile_bk
    lib = L.load()
    from smap_amd.engine import TAIL_BN
    for t in range(0, 100):
        assert lib.smap_conv_tile_tail_bn(t) == TAIL_BN.get(t, 0), t
        bm, bn = C.c_int(), C.c_int()
        rc = lib.smap_conv_tile_dims(t, C.byref(bm), C.byref(bn))
        assert (rc == 0) == (t in TILES), t
        if rc == 0:
            assert (bm.value, bn.value) == TILES[t], t
            for prec in (0, 1):
                assert lib.smap_conv_tile_bk(t, prec) == tile_bk(t, bool(prec)), (t, prec)
        else:
            assert lib.smap_conv_tile_bk(t, 0) == 0


def test_weight_packing_is_a_permutation_and_inverts():
    from smap_amd.engine import TILES, pack_conv_weights, unpack_conv_weights
    for tile, x3, k, cin, cout_pad in ((0, False, 3, 128, 256), (20, True, 1, 64, 128), (31, True, 3, 64, 256), (36, False, 3, 64, 64),
                                       (52, True, 1, 128, 256), (60, True, 3, 64, 512), (3, True, 3, 64, 32)):
        planes, K = (2 if x3 else 1), k * k * cin
        w = torch.randn(planes, cout_pad, K).half()
        p = pack_conv_weights(w, tile, x3, k, cin)
        assert p.numel() == w.numel() and torch.equal(p.reshape(-1).sort().values, w.reshape(-1).sort().values)
        assert torch.equal(unpack_conv_weights(p.reshape(-1), tile, x3, k, cin, cout_pad), w)


def test_weight_packing_matches_the_kernels_address_arithmetic():
    """pack_conv_weights against the byte addresses the kernels compute (csrc/conv.hip issue_b, conv3.hip, convp.hip): element
    (plane, n, k) must sit where lane (row, slot) of the LDS-DMA that stages its K tile reads it."""
    import random
    from smap_amd.engine import TILES, pack_conv_weights, tile_bk, tile_family
    random.seed(0)
    for tile, x3, ks, cin, cout_pad in ((0, False, 3, 128, 256), (0, True, 1, 256, 256), (20, True, 3, 64, 128), (22, True, 1, 256, 256),
                                        (52, True, 1, 128, 256), (31, True, 3, 128, 256), (31, False, 3, 128, 256), (36, True, 3, 64, 64),
                                        (38, False, 3, 256, 32), (60, True, 1, 64, 512), (62, False, 3, 64, 256), (64, True, 1, 128, 64)):
        planes, K, bn = (2 if x3 else 1), ks * ks * cin, TILES[tile][1]
        w2 = torch.arange(planes * cout_pad * K, dtype=torch.float32).reshape(planes, cout_pad, K)
        for pairs in (True, False):
          P = pack_conv_weights(w2, tile, x3, ks, cin, pairs=pairs).reshape(-1)
          for _ in range(300):
              n = random.randrange(cout_pad)
              nt, r = n // bn, n % bn
              if tile_family(tile) == "halo":
                  ch = 32 if x3 else 64
                  cch = cin // ch
                  cc, tap, s, e = random.randrange(cch), random.randrange(9), random.randrange(8), random.randrange(8)
                  gl = s ^ ((r >> 1) & 7)
                  pl, g = ((gl >> 2), (gl & 3)) if x3 else (0, gl)
                  k = tap * cin + cc * ch + g * 8 + e
                  off = ((((nt * cch + cc) * 9 + tap) * bn + r) * 8 + s) * 8 + e
              else:
                  bk = tile_bk(tile, x3)
                  spr, KT = bk // 8, K // bk
                  it, pl, s, e = random.randrange(KT), random.randrange(planes), random.randrange(spr), random.randrange(8)
                  g = s ^ (((r >> 1) & 7) if bk == 64 else ((r >> 2) & 3))
                  k = it * bk + g * 8 + e
                  if bk == 64 or not pairs:        # byte = (nt*KT + it)*WBLK + (pl*BN + r)*ROWB + s*16
                      off = ((((nt * KT + it) * planes + pl) * bn + r) * spr + s) * 8 + e
                  else:               # pairs: byte = (nt*KT/2 + it/2)*WBLK + (it&1)*64 + (pl*BN + r)*128 + s*16
                      off = (((((nt * (KT // 2) + it // 2) * planes + pl) * bn + r) * 2 + (it & 1)) * 4 + s) * 8 + e
              assert P[off] == w2[pl, n, k], (tile, x3, off)


def test_fused_bottleneck_tail_schedule_on_cpu(small_sd, monkeypatch):
    """SMAP_TAIL routes stride-1 Bottlenecks to the one-launch 3x3 + 1x1 op (csrc/convf.hip): the op list shrinks by one launch
    per such block, the torch interpretation of the fused schedule equals the unfused one bit for bit in both its modes (the
    quantised mode reads the weights back out of the packed blob: the 1x1's chunk blocks invert), the plan accepts the ops
    and rejects inconsistent ones."""
    import ctypes as C
    from oracle.graph_interp import run_graph
    from smap_amd import lib as L
    from smap_amd.engine import Graph, OP_CONV, TAIL_BN, pack_halo_rows, unpack_halo_rows
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(3))
    monkeypatch.setenv("SMAP_CAT", "0")                  # (this test compares ROUNDINGS op for op: the shortcut tensors stay stored on both sides)
    g0 = Graph(small_sd, 2, 64, 96, keep_ref=True)
    want, want_q = run_graph(g0, x.double(), quantize=False), run_graph(g0, x, quantize=True)
    monkeypatch.setenv("SMAP_TAIL", "64:80,128:82")
    monkeypatch.setenv("SMAP_BLOCK", "")                 # (the whole-block launches would take layer1's blocks in split precision)
    monkeypatch.setenv("SMAP_BLOCK_FIRST", "")
    for prec in ("f16", "x3"):
        g = Graph(small_sd, 2, 64, 96, keep_ref=True, precision=prec)
        tails = [op for op in g.ops if op.kind == OP_CONV and "tail" in op.p]
        assert len(tails) == 18 and len(g.ops) == len(g0.ops) - 18            # layer1: 3 blocks, layer2: 3 stride-1 blocks, x 3 stages
        assert all(op.res is not None and op.p["Cout"] * 4 == op.p["tail"]["cout"] for op in tails)
        assert all(torch.equal(a, b) for a, b in zip(run_graph(g, x.double(), quantize=False), want))
        if prec == "f16":
            assert all(torch.equal(a, b) for a, b in zip(run_graph(g, x, quantize=True), want_q))
        g.allocate()
        ops = g.emit()
        h = C.c_void_p()
        assert L.load().smap_plan_create(ops, len(g.ops), C.byref(h)) == 0
        L.load().smap_plan_destroy(h)
        i = next(k for k, op in enumerate(g.ops) if op.kind == OP_CONV and "tail" in op.p)
        for field, bad in (("tail_cout", 0), ("tail_cout", 100), ("tail_cout_pad", 32), ("stride", 2), ("tile", 36), ("tail_w_off", -1)):
            keep = getattr(ops[i], field)
            setattr(ops[i], field, bad)
            assert L.load().smap_plan_create(ops, len(g.ops), C.byref(h)) != 0, field
            setattr(ops[i], field, keep)
    for x3 in (False, True):
        w = torch.randn(2 if x3 else 1, 256, 64).half()
        p = pack_halo_rows(w, TAIL_BN[80], 1, 64, x3)
        assert p.shape[:3] == (4, 2 if x3 else 1, 1) and torch.equal(unpack_halo_rows(p.reshape(-1), 64, 1, 64, 256, x3), w)


def test_lazy_records_equal_eager_records():
    from smap_amd.records import frame_record, to_jsonable, train_records
    rng = np.random.default_rng(0)
    p2, p3, rz = rng.normal(size=(3, 15, 4)).astype(np.float32), rng.normal(size=(3, 15, 4)), rng.normal(size=3)
    gt = rng.normal(size=(3, 15, 11))
    for g in (None, gt):
        eager = frame_record(p2, p3, rz, "a.jpg", g)
        lazy = frame_record(p2, p3, rz, "a.jpg", g, as_lists=False)
        p2_before = p2.copy()
        assert isinstance(lazy["pred_2d"], np.ndarray) and lazy["pred_2d"] is not p2
        assert to_jsonable([lazy]) == [eager] and json.dumps(to_jsonable([lazy])) == json.dumps([eager])
        assert np.array_equal(p2, p2_before)
    assert to_jsonable(train_records(p2, p3, rz, gt, as_lists=False)) == train_records(p2, p3, rz, gt)


def test_parity_compare_counts_what_it_says():
    from benchkit import parity

    def frame(shift=0.0, drop_person=False, z_err=0.0):
        peaks = np.zeros((15, 128, 3), np.float32)
        for c in range(15):
            peaks[c, 0, 0] = 4
            peaks[c, 1:5, :2] = np.array([[10, 10], [50, 20], [90, 60], [150, 100]]) + c + shift
            peaks[c, 1:5, 2] = 0.9
        bodys = np.zeros((2, 15, 4), np.float32)
        bodys[:, :, :2] = np.arange(60).reshape(2, 15, 2) + shift
        bodys[:, :, 3] = 1.0
        p3 = np.zeros((2, 15, 4))
        p3[:, :, :3] = np.arange(90).reshape(2, 15, 3) + z_err
        p3[:, :, 3] = 1.0
        n = 1 if drop_person else 2
        maps = {k: np.ones((2, 4, 4), np.float32) for k in ("hms", "det_d", "root_d")}
        return dict(peaks=peaks, bodys=bodys[:n], p2=bodys[:n] * 4, p3=p3[:n], rz=np.array([300.0, 310.0])[:n] + z_err, **maps)

    same = parity.compare([frame()], [frame()])
    assert same["peak_match"] == same["person_match"] == same["limb_match"] == 1.0 and same["max_joint_err_cm"] == 0.0
    assert same["peaks_differing"] == 0 and same["peaks_clear_mismatch"] == 0
    assert same["peaks_ref"] == 60 and same["persons_ref"] == 2 and same["joints_compared"] == 30
    sub = parity.compare([frame(shift=0.2, z_err=0.05)], [frame()])           # sub-pixel shift: still the same peaks
    assert sub["peak_match"] == 1.0 and sub["limb_match"] == 1.0
    assert abs(sub["max_joint_err_cm"] - 0.05 * 3 ** 0.5) < 1e-9 and abs(sub["root_z_max_err_cm"] - 0.05) < 1e-9
    moved = parity.compare([frame(shift=1.0)], [frame()])                     # one pixel away: different peaks
    assert moved["peak_match"] < 0.5 and moved["person_match"] == 0.0
    lost = parity.compare([frame(drop_person=True)], [frame()])
    assert lost["person_match"] == 0.5 and lost["peak_match"] == 1.0


def test_lifter_ties_are_told_from_real_joint_errors():
    """generate_relZ samples the depth maps at ROUNDED positions (test_util.py:66,74-79): a coordinate 1e-6 px from an index
    step puts one sample on the neighbouring depth pixel -- a joint error of centimetres from a 1e-6 px input difference.
    benchkit/parity.py must call that a lifter tie, and must NOT excuse the same error when the coordinates differ by more."""
    from benchkit import parity
    from oracle import oracle_lib as O
    rng = np.random.default_rng(3)
    H, W = 32, 48
    det_d = rng.normal(0, 30, (14, H, W)).astype(np.float32)          # rough depth maps: a moved sample is visible
    root_d = rng.uniform(0.5, 1.0, (H, W)).astype(np.float32)
    cam = np.asarray([1.0, 4 * W, 4 * H, 4 * W, 4 * H, 1000.0, 1000.0, 2 * W, 2 * H], np.float64)
    body = np.zeros((1, 15, 4), np.float32)
    body[0, :, 0] = rng.uniform(8, W - 8, 15)
    body[0, :, 1] = rng.uniform(8, H - 8, 15)
    body[0, :, 3] = 1.0
    body[0, 0, :2] = (20.25, 10.25)                                   # pelvis and neck well inside a depth pixel
    body[0, 2, :2] = (24.25, 14.25)

    def frame(bd):
        p2, p3, rz = O.lift(bd, det_d, root_d, cam)
        peaks = np.zeros((15, 128, 3), np.float32)
        return dict(peaks=peaks, bodys=bd, p2=p2, p3=p3, rz=rz, det_d=det_d, root_d=root_d, hms=np.ones((43, H, W), np.float32))

    # limb 8 = (2, 12): put joint 12 so that its x4 coordinate sits on the step between two depth pixels: round(x) = 4m - 0.5
    step = np.float32((4 * 30 - 0.5) / 4)                             # x4 -> 119.5 exactly: np.round goes to 120 (even) = pixel 30
    a, b = body.copy(), body.copy()
    a[0, 12, 0] = step
    b[0, 12, 0] = np.nextafter(step, np.float32(0))                   # one ulp below: rounds to 119 = pixel 29
    assert parity.lift_sample_pixels(a[0])[2][8][0][-1, 1] != parity.lift_sample_pixels(b[0])[2][8][0][-1, 1]
    m = parity.compare([frame(a)], [frame(b)])
    assert m["person_match"] == 1.0 and m["limb_match"] == 1.0
    assert m["max_joint_err_cm"] > 0.1, "the moved sample must matter in this scene"
    assert m["joints_over_0.1cm_unexplained"] == 0 and m["lifter_ties"] >= 1, m
    assert m["lifter_tie_max_coord_diff_px"] < 5e-4
    # the same sample moved by a coordinate that differs by 0.2 px (still 'the same peak' for the 0.5 px pairing): not a tie
    c = body.copy()
    c[0, 12, 0] = step - np.float32(0.05)                              # 0.2 network px away: far beyond LIFT_TIE_PX
    m2 = parity.compare([frame(a)], [frame(c)])
    assert m2["limb_match"] == 1.0 and m2["max_joint_err_cm"] > 0.1
    assert m2["joints_over_0.1cm_unexplained"] >= 1 and m2["lifter_ties"] == 0, m2


def test_peak_ties_are_told_from_real_mismatches():
    """A candidate that clears the threshold by 1e-7 of the map scale in one path and misses it in the other is a floating-
    point tie; one that differs by 1e-2 is a mismatch."""
    from benchkit import parity
    rng = np.random.default_rng(5)
    base = rng.uniform(0.0, 0.1, (43, 32, 48)).astype(np.float32)
    base[3, 10, 20], base[7, 20, 30] = 0.9, 0.5                     # two clear peaks (scale = 0.9)

    def frame(hms):
        peaks = np.zeros((15, 128, 3), np.float32)
        return dict(peaks=peaks, bodys=np.zeros((0, 15, 4), np.float32), p2=np.zeros((0, 15, 4)), p3=np.zeros((0, 15, 4)),
                    rz=np.zeros((0,)), hms=hms, det_d=np.ones((1, 2, 2), np.float32), root_d=np.ones((2, 2), np.float32))

    assert parity.peak_pixels(base[:15]) == {(3, 10, 20), (7, 20, 30)}
    tie_ref, tie_hip = base.copy(), base.copy()
    tie_ref[5, 8, 8], tie_hip[5, 8, 8] = 0.2 + 2e-7, 0.2 - 2e-7      # just above / just below the 0.2 threshold
    m = parity.compare([frame(tie_hip)], [frame(tie_ref)])
    assert m["peaks_differing"] == 1 and m["peaks_clear_mismatch"] == 0 and m["peaks_differing_max_margin"] < 1e-6
    bad_hip = base.copy()
    bad_hip[7, 20, 30] = 0.1                                          # a clear peak lost
    m = parity.compare([frame(bad_hip)], [frame(base)])
    assert m["peaks_differing"] == 1 and m["peaks_clear_mismatch"] == 1 and m["peaks_differing_max_margin"] > 0.1
    # more than 127 candidates in a channel: a tie near the top of the raster also moves the cap at its end -- one decision
    # differs (the tie), one perfectly clear peak is swapped in / out as a CONSEQUENCE (counted apart, not a mismatch)
    crowd = base.copy()
    ys, xs = np.meshgrid(np.arange(4, 30, 2), np.arange(2, 46, 2), indexing="ij")
    crowd[9, ys.ravel()[:130], xs.ravel()[:130]] = 0.9                 # 130 isolated clear peaks in channel 9 (rows 4..)
    c_ref, c_hip = crowd.copy(), crowd.copy()
    c_ref[9, 1, 8], c_hip[9, 1, 8] = 0.2 + 2e-7, 0.2 - 2e-7           # a tie above them all in raster order
    m = parity.compare([frame(c_hip)], [frame(c_ref)])
    assert m["peaks_differing"] == 1 and m["peaks_clear_mismatch"] == 0 and m["peaks_cap_shifted"] == 1


def test_people_weights_are_calibrated_data_not_weights():
    from benchkit.workload import HEAD_KINDS, people_state_dict
    cal = json.load(open(os.path.join(ROOT, "benchkit", "head_calibration.json")))
    assert set(cal) == set(HEAD_KINDS) and all(len(cal[k]["kpt_bias"]) == 15 for k in cal)
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    base = SMAP(make_cfg((16, 24))).state_dict()
    a, b = recipe_state_dict(base), people_state_dict(base, "smooth")
    changed = sorted(k for k in a if not torch.equal(a[k], b[k]))
    assert changed == ["stage2.upsample.up4.res_conv2.bn.bias", "stage2.upsample.up4.res_conv2.bn.weight",
                       "stage2.upsample.up4.res_rd_conv2.bn.bias", "stage2.upsample.up4.res_rd_conv2.bn.weight"]


# ------------------------------------------------------------------ round 5: merged launches, split K, small-schedule rules, lanes
def _full_size_sd():
    from types import SimpleNamespace as NS
    from smap_amd.model.smap import SMAP
    cfg = NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256), DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
             OUTPUT_SHAPE=(128, 208), LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))
    torch.manual_seed(0)
    return SMAP(cfg).state_dict()


def test_batch_1_schedule_rules_and_arena(monkeypatch):
    """What the batch-1 schedule (BASELINE configs[1]) is made of, without a GPU: split K only on the long-K launches of the coarse
    levels (K >= 2048, <= 4 parts, workgroups x parts <= 256 there), the four-stage 64x64 tile exactly on the launches of <= 256
    workgroups, layer1's whole blocks on 4x16 tiles, layer2's as three launches; every split op has its own scratch and ticket slice,
    scratch and tickets never share bytes with a live tensor; the library accepts the plan; with lanes on, every wait names an earlier
    op of another lane and nothing a side-lane op touches is reused before the end of the schedule."""
    import ctypes as C
    from smap_amd import lib as L
    from smap_amd.engine import Graph, OP_CONV, TILES, tile_bk
    sd = _full_size_sd()
    for k in ("SMAP_CAT", "SMAP_SKIPSUM", "SMAP_TAPHEAD"):          # (conftest forces round 6's launches on for the small TEST schedules; this is the real batch-1 rule)
        monkeypatch.delenv(k, raising=False)
    for lanes in ("0", "1"):
        monkeypatch.setenv("SMAP_LANES", lanes)
        g = Graph(sd, 1, 512, 832, precision="x3")
        g.allocate()
        ops = g.emit()
        convs = [(op, o) for op, o in zip(g.ops, ops) if op.kind == OP_CONV]
        assert sorted({op.p["tile"] for op, _ in convs if op.p["tile"] >= 90}) == [90, 92]              # layer1: 4 x 16 tiles; no tile 94
        split = [(op, o) for op, o in convs if o.ksplit > 1]
        assert 30 <= len(split) <= 40
        tickets = set()
        for op, o in split:
            K = o.ksize * o.ksize * o.Cin
            bm, bn = TILES[o.tile]
            tiles = -(-(o.B * o.Ho * o.Wo) // bm) * (o.cout_pad // bn)
            assert K >= 2048 and 2 <= o.ksplit <= 4 and tiles * o.ksplit <= 256 and o.ksplit <= K // tile_bk(o.tile, True)
            assert o.kpart_off == op.scratch[0].off and op.scratch[0].nbytes == tiles * o.ksplit * bm * bn * 4
            lo = o.kcount_off
            assert g.kcount.off <= lo and lo + 4 * tiles <= g.kcount.off + g.kcount.nbytes
            assert not tickets & set(range(lo, lo + 4 * tiles, 4))
            tickets |= set(range(lo, lo + 4 * tiles, 4))
        for op, o in convs:                                        # the deep-pipeline rule
            if o.tile in (2, 7):
                wgs = -(-(o.B * o.Ho * o.Wo) // 64) * (o.cout_pad // 64) * max(1, o.ksplit)
                assert (o.tile == 7) == (wgs <= 256), (op.out.name, wgs)
        every = g.tensors + g.scratch_tensors + [g.kcount]
        live = sorted((t.first, t.last, t.off, t.off + t.nbytes, t.name) for t in every)
        for i, a in enumerate(live):
            for b in live[i + 1:]:
                if b[0] > a[1]:
                    break
                assert a[3] <= b[2] or b[3] <= a[2], (a[4], b[4])
        n = len(g.ops)
        if lanes == "1":
            assert sorted({op.lane for op in g.ops}) == [0, 1, 2]
            for i, (op, o) in enumerate(zip(g.ops, ops)):
                for k in range(o.n_wait):
                    assert 0 <= o.wait_op[k] < i and ops[o.wait_op[k]].lane != o.lane
                if op.lane:
                    for t in [op.inp, op.res, op.add1, op.add2] + list(op.aux):
                        assert t is None or t.last == n - 1
        else:
            assert all(o.lane == 0 and o.n_wait == 0 for o in ops)
        h = C.c_void_p()
        assert L.load().smap_plan_create(ops, n, C.byref(h)) == 0
        rc = L.load().smap_plan_set_lanes(h, int(lanes))          # lanes on: creates the side streams IN this call (include/smap_hip.h) --
        assert rc == 0 or (lanes == "1" and not torch.cuda.is_available() and rc <= -1000), rc     # a hipError_t on a host without a GPU
        if rc == 0:
            assert L.load().smap_plan_set_lanes(h, 0) == 0
        L.load().smap_plan_destroy(h)


def test_merged_launch_descriptors_and_validation(small_sd, monkeypatch):
    """Graph.conv_seg -> smap_op.seg_*: segment starts are multiples of the tile's N extent and in order, every segment has its own
    output tensor / scale, the plan validates -- and refuses a segment that is mis-aligned, overlaps the next one, or sits on a tile
    family without the per-segment epilogue."""
    import ctypes as C
    from smap_amd import lib as L
    from smap_amd.engine import Graph, OP_CONV, TILES
    monkeypatch.setenv("SMAP_SKIPSUM", "0")              # (the skip convs as segments: with the default skip1 + skip2 launches only stage 2's two merged launches remain)
    for merge, n in (("2", 18), ("1", 12), ("0", 0)):
        monkeypatch.setenv("SMAP_MERGE_1X1", merge)
        g = Graph(small_sd, 2, 64, 96, precision="x3")
        g.allocate()
        ops = g.emit()
        segd = [(op, o) for op, o in zip(g.ops, ops) if op.kind == OP_CONV and o.seg_n[0] > 0]
        assert len(segd) == n
        for op, o in segd:
            bn = TILES[o.tile][1]
            starts = [s for s in o.seg_n if s > 0]
            assert starts == sorted(starts) and all(s % bn == 0 for s in starts) and o.Cout <= starts[0] and o.ksize == 1
            for j, t in enumerate(op.outs):
                assert o.seg_out_off[j] == t.off and o.seg_cout[j] == t.C and o.seg_out_stride_c[j] == 2 * t.C and o.seg_acc_scale[j] > 0
            assert len({op.out.off} | {t.off for t in op.outs}) == 1 + len(op.outs)
        h = C.c_void_p()
        assert L.load().smap_plan_create(ops, len(g.ops), C.byref(h)) == 0
        L.load().smap_plan_destroy(h)
        if segd:
            i = next(k for k, op in enumerate(g.ops) if op.outs)
            for field, val in (("seg_n", ops[i].seg_n[0] + 8), ("seg_cout", ops[i].seg_cout[0] + 4), ("tile", 60), ("ksize", 3)):
                bad = (L.SmapOp * len(g.ops))()
                C.memmove(bad, ops, C.sizeof(bad))
                if field in ("seg_n", "seg_cout"):
                    getattr(bad[i], field)[0] = val
                else:
                    setattr(bad[i], field, val)
                assert L.load().smap_plan_create(bad, len(g.ops), C.byref(h)) != 0, field
