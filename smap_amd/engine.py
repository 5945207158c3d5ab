"""Host side of the SMAP backbone on MI355X: folds the checkpoint, lays the static
inference schedule out as an array of `smap_op` (include/smap_hip.h) and runs it
through libsmap_hip.so.  Python only DESCRIBES the schedule; every kernel launch
happens inside smap_plan_run (csrc/plan.hip).

Reference being replaced: model/smap.py SMAP.forward, inference branch (:403-419).

What the schedule does differently from a layer-by-layer replay (all exact up to
floating-point rounding):
  * eval-mode BatchNorm is folded into the conv weights / bias (smap.py:13-45);
  * heads that the inference branch never returns are not computed (smap.py:417-419 uses
    only stage-2 res2..4, res_d4, res_rd4): 62 of the 268 convs;
  * `up_conv(bilinear_up(x))` (smap.py:214-215) is evaluated as `bilinear_up(up_conv(x))`:
    a 1x1 conv + per-channel affine commutes with the (convex, per-channel) bilinear
    resampling, which moves the 256->256 GEMM to the 4x smaller grid;
  * the three 1x1 head convs of stage-2/up4 share their input and run as one N=768 GEMM;
  * the 1x1 convs of an Upsample_unit that read the same tensor (smap.py:210-241: skip2 | cross_conv / res_conv1 | the next unit's
    up_conv on `out`; u_skip | skip1 on x where u_skip has no bilinear add) run as ONE launch with one output tensor each (conv_seg);
  * layer1's Bottlenecks and layer2's identity Bottlenecks run as one launch each (conv_block*, csrc/convb.hip / convc.hip);
  * small schedules (batch 1): the long-K launches split their K loop over several workgroups per output tile (split_k);
  * residual add, ReLU and the inter-stage skip adds (smap.py:142-153) run in the
    epilogue of the producing conv.
Activations are NHWC fp16, accumulation fp32, head outputs fp32 (precision "f16"), or -- precision "x3", the mode
whose results meet the reference's fp32 arithmetic -- every activation and weight as an fp16 hi/lo PAIR (22 significant
bits) with three MFMAs per K step (csrc/conv.hip, template X3): ~1e-6 relative error through the whole graph instead of
~1e-3, at 3x the matrix work and 2x the bytes.
"""
import ctypes as C
from dataclasses import dataclass, field

import torch

from . import lib as _L

OP_CONV, OP_STEM, OP_MAXPOOL, OP_UPADD, OP_HEADSUM, OP_STEMPOOL, OP_TAPSUM = range(7)
import os

# tile id -> (BM, BN); ids 5..9 are the same tiles with deeper LDS-DMA pipelines (csrc/conv.hip)
TILES = {0: (128, 128), 1: (128, 64), 2: (64, 64), 3: (128, 32), 4: (64, 128),
         5: (128, 128), 6: (128, 64), 7: (64, 64), 8: (128, 32), 9: (64, 128),
         # 20..27: csrc/conv.hip with BK = 32 staging (LDS-staged epilogue kept)
         20: (128, 128), 21: (128, 64), 22: (64, 64), 23: (64, 128), 24: (128, 128), 25: (128, 64), 26: (64, 64),
         27: (64, 128),
         # 30..39: csrc/conv3.hip halo-tiled 3x3 stride-1 (BM = 128 output pixels as an 8x16 / 4x32 patch)
         30: (128, 64), 31: (128, 128), 32: (128, 64), 33: (128, 128),
         34: (128, 64), 35: (128, 128), 36: (128, 64), 37: (128, 128),     # 34..37: deeper weight pipeline
         38: (128, 32), 39: (128, 32),                                     # Cout <= 32 heads
         # 40..43: the same kernel with EIGHT waves: 8x16 pixels (two workgroups per CU), 16x16 / 16x16 x 64 channels / 8x32 (one)
         40: (128, 128), 41: (256, 128), 42: (256, 64), 43: (256, 128),
         44: (256, 128), 45: (256, 128),                                   # 41 / 43 with the two waves of a SIMD half an iteration apart
         # 50..54: csrc/conv.hip with EIGHT waves per workgroup (two per SIMD from one workgroup: the low-resolution layers)
         50: (128, 128), 51: (128, 128), 52: (128, 128), 53: (256, 128), 54: (128, 256),
         # 55: 4-stage pipeline (three K tiles in flight per workgroup: bytes in flight, not occupancy, for the streaming layers)
         55: (128, 128),
         # 56 (round 6): 256 x 256, eight waves of 128 x 64, REGISTER epilogue (two 64 KiB LDS stages leave no room for a staging tile): half the
         # L2 -> LDS bytes per MFMA of the 128 x 128 tiles.  Split precision, fp16 outputs, residual + ReLU only (no bilinear add / addends)
         56: (256, 256),
         # 60..62: csrc/convp.hip, persistent workgroups with loader waves and a register epilogue (no fused bilinear add, no fp32 out)
         60: (128, 256), 61: (256, 128), 62: (128, 128), 63: (128, 64), 64: (128, 64), 65: (128, 64),
         # 80..82: csrc/convf.hip, a Bottleneck's 3x3 (BN = all of its planes) with the following 1x1 fused in (TAIL_BN)
         80: (128, 64), 81: (128, 64), 82: (128, 128),
         # 90..91: csrc/convb.hip, a whole identity Bottleneck of 64 planes (1x1 -> 3x3 -> 1x1 + residual) per 4x16 / 8x16 pixel tile
         90: (64, 64), 91: (128, 64),
         # 92..93: the same for the FIRST block of layer1 (64 input channels, 1x1 shortcut conv instead of the identity residual)
         92: (64, 64), 93: (128, 64),
         # 94: csrc/convc.hip, the whole identity Bottleneck of 128 planes / 512 channels (layer2) per 8x16 pixel tile, eight waves
         94: (128, 128)}
TAIL_DEFAULT = {}                          # Bottleneck planes -> fused tile id (empty: every block runs c2 and c3 as two launches)
TAIL_BN = {80: 64, 81: 128, 82: 64, 90: 64, 91: 64, 92: 64, 93: 64, 94: 128}      # output channels per chunk of the fused 1x1 (csrc/convf.hip::smap_convf_tile_dims)
# Bottleneck planes -> tile id of the WHOLE-block launch (csrc/convb.hip) for stride-1 identity blocks in split precision; {} = off.
# SMAP_BLOCK="64:91" overrides (A/B hook; "" = off).  BLOCK_FIRST_DEFAULT / SMAP_BLOCK_FIRST="64:93": the same for the first block of
# layer1 (the one with a shortcut conv; 64 input channels).
# Measured in situ (profiles/r4_v4_ab_whole_block_first.log, same box, interleaved): 781 -> 800 (identity blocks, 8 x 16 tiles) -> 821
# frames/s (+ first blocks); 4 x 16 tiles: 816.
BLOCK_DEFAULT = {64: 91, 128: 94}       # layer1 (csrc/convb.hip) and layer2 (csrc/convc.hip) identity blocks
BLOCK_FIRST_DEFAULT = {64: 93}
# (ids 10..18 belonged to a register-epilogue GEMM that no measured table entry selected; it lives on as an experiment under
#  tools/experiments/, outside the product build.  ids 40..45 are live: the eight-wave halo tiles of csrc/conv3.hip.)


def _tile_remap():
    """SMAP_TILE_REMAP="0:5,1:6" swaps tile variants without touching the schedule (A/B runs)."""
    out = {}
    for kv in os.environ.get("SMAP_TILE_REMAP", "").split(","):
        if ":" in kv:
            a, b = kv.split(":")
            assert TILES[int(a)] == TILES[int(b)], "remap must keep the tile shape"
            out[int(a)] = int(b)
    return out
LAYERS = (3, 4, 6, 3)            # smap.py:299  resnet-50
PLANES = (64, 128, 256, 512)
ALIGN = 256
ZERO_PAGE = 16384             # csrc/plan.hip SMAP_ZERO_PAGE
WINDOW = 1 << 32              # csrc/plan.hip SMAP_WINDOW: bytes [k * WINDOW, k * WINDOW + ZERO_PAGE) of the arena are reserved
PRECISIONS = ("f16", "x3")
SPLITK_TILES = (2, 7, 20, 22)     # csrc/conv.hip tiles with a split-K instance (smap_conv_tile_has_splitk)
X3_TILES = (0, 1, 2, 4, 7, 20, 21, 22, 23, 24, 25, 26, 27, 50, 51, 52, 53, 54, 55, 56, 60, 61, 62, 63, 64, 65)   # conv.hip tiles with a split-precision instance (+ 3: Cout <= 32)


def split_f16(w, scaled=True):
    """f64 tensor -> (hi, lo fp16 tensors of w * 2^s, 2^-s): hi = fp16(w 2^s), lo = fp16(w 2^s - hi).  s puts max |w| in
    [2^13, 2^14) so that the lo parts of all but vanishing weights are normal fp16 numbers."""
    import math
    m = float(w.abs().max())
    s = (13 - math.frexp(m)[1] + 1) if (scaled and m > 0) else 0          # frexp: m = f * 2^e, f in [0.5, 1)
    ws = w.double() * (2.0 ** s)
    hi = ws.to(torch.float16)
    lo = (ws - hi.double()).to(torch.float16)
    assert torch.isfinite(hi).all()
    return hi, lo, 2.0 ** -s


_TILE_TABLE_X3 = None


def _table_entry(v):
    """A table value is one tile id or a ranked list of them (best first): the specialised kernels do not take every op of
    a shape (conv3.hip: plain 3x3 only; convp.hip: no fused bilinear add, no fp32 output), Graph.conv picks the first
    entry that is legal for the op at hand."""
    return [int(t) for t in v] if isinstance(v, (list, tuple)) else [int(v)]


REGEPI_TILES = (56,)               # csrc/conv.hip tiles with the register epilogue
DUAL_TILES = (20, 50, 51, 53, 54)  # csrc/conv.hip tiles with a second-input (K-concatenated) instance (smap_conv_tile_has_dual)
RELUSUM_TILES = (50, 51, 53, 54)   # ... with a relu(conv) + relu(conv) instance (smap_conv_tile_has_relusum)


def tile_legal(tile, *, cout, cout_pad=None, plain3=True, up=False, out_fp32=False, adds=False, x3=True):
    """Can tile id `tile` run an op with these properties?  Mirrors csrc/plan.hip::validate."""
    if tile in REGEPI_TILES:
        return x3 and not up and not out_fp32 and not adds and cout % 8 == 0
    if 30 <= tile < 50:
        return plain3
    if 80 <= tile < 100:
        return False                     # only Graph.conv_tail / Graph.conv_block build these
    if 60 <= tile < 80:
        cp = cout_pad if cout_pad is not None else _rup(cout, TILES[tile][1])
        return not up and not out_fp32 and cout % 8 == 0 and cp <= 2048
    return True


def pick_tile_x3(M, cout, key=None):
    """Split precision: candidates in order of preference -- the measured table (tools/autotune.py --precision x3 ->
    smap_amd/tile_table_x3.json) when the shape is in it, then the largest BK = 32 tile that still gives >= 512
    workgroups / the one with most workgroups."""
    global _TILE_TABLE_X3
    if _TILE_TABLE_X3 is None:
        import json
        path = os.environ.get("SMAP_TILE_TABLE_X3") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_table_x3.json")
        _TILE_TABLE_X3 = json.load(open(path)) if os.path.exists(path) and not os.environ.get("SMAP_NO_TILE_TABLE") else {}
    keys = [key] if isinstance(key, str) else list(key or [])          # several keys: most specific first
    # a batch size the table was not tuned for borrows the entries of the nearest tuned one (same layer shape, M within 2x-4x):
    # closer to the optimum than the block-count heuristic (tuned tables: +8 % at batch 8, +7 % at 16, +19 % at 1)
    tuned = sorted({int(k.split(",")[0]) for k in _TILE_TABLE_X3})
    if keys and tuned:
        f = int(keys[0].split(",")[0])
        if f not in tuned:
            import math
            near = min(tuned, key=lambda b: abs(math.log(b / f)))
            keys = keys + [",".join([str(near)] + k.split(",")[1:]) for k in keys]
    cands = [t for k in keys if k in _TILE_TABLE_X3 for t in _table_entry(_TILE_TABLE_X3[k])]
    if cout <= 32:
        return cands + [3]
    best, best_blocks = None, -1
    for t in ((21, 22) if cout <= 64 else (20, 23, 21, 22)):
        bm, bn = TILES[t]
        blocks = -(-M // bm) * (-(-cout // bn))
        if blocks >= 512:
            return cands + [t]
        if blocks > best_blocks:
            best, best_blocks = t, blocks
    return cands + [best]


def tile_family(tile):
    """Which kernel a tile id selects: "halo" (csrc/conv3.hip; csrc/convf.hip = the same 3x3 with a fused 1x1 tail),
    "persist" (csrc/convp.hip) or "igemm" (csrc/conv.hip)."""
    return "halo" if (30 <= tile < 50 or 80 <= tile < 100) else "persist" if 60 <= tile < 80 else "igemm"


def tile_bk(tile, x3):
    """Halves per K chunk of the LDS rows a tile id stages (csrc/conv.hip::smap_launch_conv, csrc/convp.hip; the C side exports
    the same table as smap_conv_tile_bk, tests/test_host_cpu.py keeps the two in step).  The halo kernel's rows are always
    128 bytes: 64 channels, or [hi32 | lo32] of 32 channels in split precision."""
    fam = tile_family(tile)
    if fam == "halo":
        return 32 if x3 else 64
    if fam == "persist":
        return 32
    if x3:
        return 64 if tile in (0, 1, 2, 3, 4, 7, 52) else 32
    return 32 if tile in (20, 21, 22, 23, 24, 25, 26, 27, 50, 51, 53, 54, 55) else 64


# 32-half K tiles of the packed weights are stored in PAIRS (128-byte rows [tile 2p | tile 2p+1]) for SMALL schedules (<= 2 frames of
# 512x832 worth of pixels): every launch of those is latency-bound and lives on the L1 hit the second half of each fetched line
# gives the next K tile (batch 1: 226 vs 195 frames/s); larger schedules are bandwidth-bound and prefer one contiguous block
# per K tile (batch 8: 751 vs 738).  A per-launch rule (pairs for <= 512 workgroups) sits in between on both (226 / 745):
# profiles/r3_ab_wpairs*.log.  SMAP_WPAIRS=0|1 forces one layout.
def use_lanes(frames, H, W):
    """Forked streams for the independent head chains (smap_op.lane).  OFF by default: measured at batch 1 (profiles/r5_v3_ab_b1.log) the
    forward is SLOWER with them, launched kernel by kernel (3.619 -> 3.706 ms) and replayed from a graph (3.129 -> 3.436 ms): the head
    chains are 6 of 170 launches, and every fork / join costs an event round trip on the critical path.  SMAP_LANES=1 switches them on
    (the mechanism stays tested: tests/test_backbone_gpu.py::test_lanes_fork_the_head_chains_and_change_nothing)."""
    return int(os.environ.get("SMAP_LANES", "0") == "1")


def use_w_pairs(frames, H, W):
    forced = os.environ.get("SMAP_WPAIRS", "")
    return int(forced) if forced in ("0", "1") else int(frames * H * W <= 2 * 512 * 832)


def pack_conv_weights(w2, tile, x3, ksize, cin, pairs=True):
    """fp16 [planes][cout_pad][K] (K = (kh, kw, cin), planes = hi | lo in split precision) -> the byte image the conv kernels'
    LDS-DMA reads, as ONE CONTIGUOUS BLOCK PER STAGED WEIGHT TILE: [n tile][K tile][plane][row][16-byte slot], the slot
    order already carrying the kernels' XOR swizzle -- a weight tile is then BN x row-bytes of consecutive addresses, i.e.
    every wave-wide global_load_lds reads one contiguous KiB.  (Strided 64-byte row segments out of a [cout][K] matrix
    stream from L2 at half the rate: profiles/r3_v3_ubench_lds_dma_rows.log; weight tiles are re-streamed by every M tile
    and were the larger half of the L2 -> LDS traffic.)
      igemm / persist (csrc/conv.hip, convp.hip): K tile = BK halves in (kh, kw, cin) order; slot s of row r holds granule
        s ^ ((r >> 1) & 7) (BK = 64) or s ^ ((r >> 2) & 3) (BK = 32); 32-half tiles come in pairs [n tile][pair][plane][row]
        [tile 2p | tile 2p+1]: 128-byte rows again, the second half of every fetched cache line is the next K tile;
      halo (csrc/conv3.hip): tiles ordered [channel chunk][tap]; 128-byte rows of 64 channels, or in split precision of
        32 channels as logical granules 0..3 = hi, 4..7 = lo; slot s of row r holds logical granule s ^ ((r >> 1) & 7)."""
    planes, cout_pad, K = w2.shape
    assert planes == (2 if x3 else 1) and K == ksize * ksize * cin
    bn = TILES[tile][1]
    nt = cout_pad // bn
    assert nt * bn == cout_pad
    r = torch.arange(bn)
    if tile_family(tile) == "halo":
        assert ksize == 3
        return pack_halo_rows(w2, bn, 9, cin, x3)
    bk = tile_bk(tile, x3)
    spr = bk // 8
    kt = K // bk
    assert kt * bk == K
    rows = w2.reshape(planes, nt, bn, kt, spr, 8).permute(1, 3, 0, 2, 4, 5)          # [nt][kt][plane][row][granule][8]
    swz = ((r >> 1) & 7) if bk == 64 else ((r >> 2) & 3)
    idx = torch.arange(spr)[None, :] ^ swz[:, None]
    tiles = rows[:, :, :, r[:, None], idx, :]                                        # slot order of the LDS image
    if bk == 32 and pairs:
        # 32-half K tiles are stored in PAIRS: a 128-byte row = [K tile 2p (64 B) | K tile 2p+1 (64 B)], so that the cache line
        # a K tile's load brings in also serves the next K tile (small launches are latency-bound: the L1 hit matters there)
        assert kt % 2 == 0
        tiles = tiles.reshape(nt, kt // 2, 2, planes, bn, spr, 8).permute(0, 1, 3, 4, 2, 5, 6)   # [nt][pair][plane][row][half][4][8]
    return tiles.contiguous()


def pack_halo_rows(w2, bn, taps, cin, x3):
    """[planes][cout_pad][K = (tap, cin)] -> blocks [n tile][channel chunk][tap][bn rows][8 slots][8 halves] of 128-byte rows
    (csrc/conv3.hip; taps = 1: the fused 1x1 of csrc/convf.hip, bn = its chunk of output channels)."""
    planes, cout_pad, K = w2.shape
    assert K == taps * cin and cout_pad % bn == 0
    nt = cout_pad // bn
    ch = 32 if x3 else 64
    cch = cin // ch
    assert cch * ch == cin
    r = torch.arange(bn)
    w = w2.reshape(planes, nt, bn, taps, cch, ch // 8, 8)
    if x3:       # logical granule = plane * 4 + g
        rows = w.permute(1, 4, 3, 2, 0, 5, 6).reshape(nt, cch, taps, bn, 8, 8)
    else:
        rows = w[0].permute(0, 3, 2, 1, 4, 5)
    idx = torch.arange(8)[None, :] ^ ((r[:, None] >> 1) & 7)
    return rows[:, :, :, r[:, None], idx, :].contiguous()


def pack_rows16(w2):
    """[2 planes][rows][K] (hi | lo) -> [K / 16][rows][4 slots][8 halves]: the 16-channel stages of csrc/convb.hip's leading 1x1 --
    64-byte rows [hi16 | lo16], logical granule = plane * 2 + g (g = which 8 of the 16 channels), slot s of row r holds
    granule s ^ ((r >> 2) & 3)."""
    planes, rows, K = w2.shape
    assert planes == 2 and K % 16 == 0
    w = w2.reshape(2, rows, K // 16, 2, 8).permute(2, 1, 0, 3, 4).reshape(K // 16, rows, 4, 8)
    r = torch.arange(rows)
    idx = torch.arange(4)[None, :] ^ ((r[:, None] >> 2) & 3)
    return w[:, r[:, None], idx, :].contiguous()


def unpack_halo_rows(packed, bn, taps, cin, cout_pad, x3):
    planes, K = (2 if x3 else 1), taps * cin
    perm = pack_halo_rows(torch.arange(planes * cout_pad * K, dtype=torch.int64).reshape(planes, cout_pad, K), bn, taps, cin, x3).reshape(-1)
    out = torch.empty(planes * cout_pad * K, dtype=packed.dtype)
    out[perm] = packed.reshape(-1)
    return out.reshape(planes, cout_pad, K)


def unpack_conv_weights(packed, tile, x3, ksize, cin, cout_pad, pairs=True):
    """Inverse of pack_conv_weights: the flat fp16 image -> [planes][cout_pad][K] (oracle/graph_interp.py, tests)."""
    planes, K = (2 if x3 else 1), ksize * ksize * cin
    perm = pack_conv_weights(torch.arange(planes * cout_pad * K, dtype=torch.int64).reshape(planes, cout_pad, K),
                             tile, x3, ksize, cin, pairs=pairs).reshape(-1)
    out = torch.empty(planes * cout_pad * K, dtype=packed.dtype)
    out[perm] = packed.reshape(-1)
    return out.reshape(planes, cout_pad, K)


def _rup(x, m):
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- folding
def fold_conv_bn(sd, prefix, eps=1e-5):
    """conv_bn_relu (smap.py:13-45) in eval mode -> (w [Cout,Cin,kh,kw] f64, b [Cout] f64)."""
    w = sd[prefix + ".conv.weight"].double()
    b = sd[prefix + ".conv.bias"].double()
    g = sd[prefix + ".bn.weight"].double()
    beta = sd[prefix + ".bn.bias"].double()
    mu = sd[prefix + ".bn.running_mean"].double()
    var = sd[prefix + ".bn.running_var"].double()
    s = g / torch.sqrt(var + eps)
    return w * s[:, None, None, None], (b - mu) * s + beta


# ----------------------------------------------------------------------------- graph IR
@dataclass
class Tensor:
    name: str
    B: int
    H: int
    W: int
    C: int                 # channel stride of a pixel
    esize: int = 2         # bytes per element (2 = fp16, 4 = fp32)
    planes: int = 1        # 2 = split precision: a pixel is [hi(C) | lo(C)] fp16
    first: int = -1
    last: int = -1
    off: int = -1

    @property
    def nbytes(self):
        return self.B * self.H * self.W * self.C * self.esize * self.planes


@dataclass
class Op:
    kind: int
    out: Tensor = None
    inp: Tensor = None
    res: Tensor = None
    add1: Tensor = None
    add2: Tensor = None
    aux: list = field(default_factory=list)
    p: dict = field(default_factory=dict)
    lane: int = 0                                 # stream lane (smap_op.lane): 0 = the caller's stream
    outs: list = field(default_factory=list)      # further output tensors (N segments 1, 2 of a merged 1x1 launch)
    aux2: Tensor = None                           # second input, concatenated along K (Graph.conv_cat)
    scratch: list = field(default_factory=list)   # arena scratch that lives for this op only (split K: the partial tiles)


def pick_tile_heuristic(M, cout):
    """Largest tile that still gives >= 2 waves of workgroups on 256 CUs."""
    if cout <= 32:
        return 3
    cands = [1, 2] if cout <= 64 else [0, 1, 2]
    best, best_blocks = None, -1
    for t in cands:
        bm, bn = TILES[t]
        blocks = -(-M // bm) * (-(-cout // bn))
        if blocks >= 512:
            return t
        if blocks > best_blocks:
            best, best_blocks = t, blocks
    return best


_TILE_TABLE = None


def pick_tile(M, cout, key=None):
    """Measured table (tools/autotune.py -> smap_amd/tile_table.json, keyed "B,H,W,Cin,Cout,k,s")
    when the shape is in it, else the block-count heuristic."""
    global _TILE_TABLE
    if _TILE_TABLE is None:
        import json
        path = os.environ.get("SMAP_TILE_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_table.json")
        _TILE_TABLE = json.load(open(path)) if os.path.exists(path) and not os.environ.get("SMAP_NO_TILE_TABLE") else {}
    pol = os.environ.get("SMAP_TILE_POLICY", "")
    if pol.startswith("traffic"):       # experiment: minimise L2->LDS bytes subject to a minimum block count
        min_blocks = int(pol.split(":")[1]) if ":" in pol else 256
        K = 1
        if key is not None:
            _, _, _, cin, _, ks, _ = map(int, key.split(","))
            K = ks * ks * cin
        best, best_cost = None, None
        for t in ([3] if cout <= 32 else [0, 1, 2, 4]):
            bm, bn = TILES[t]
            if cout <= 64 and bn > 64:
                continue
            mt, nt = -(-M // bm), -(-cout // bn)
            cost = (mt * bm * nt + nt * bn * mt) * K
            if mt * nt < min_blocks:
                cost *= 4
            if best is None or cost < best_cost:
                best, best_cost = t, cost
        return best
    if key is not None and key in _TILE_TABLE:
        return _table_entry(_TILE_TABLE[key])[0]
    return pick_tile_heuristic(M, cout)


DEFAULT_REMAP = {}


class ArenaTooLarge(ValueError):
    """The schedule does not fit: ONE activation tensor exceeds a 4 GiB window (the conv kernels address their input with 32-bit
    byte offsets from a 4 GiB-aligned base, csrc/plan.hip: 52 frames of the widest split-precision tensor), or the whole arena
    exceeds the memory budget (BackboneEngine: SMAP_MAX_ARENA_BYTES, default 90 % of the device's memory / 4 arenas -- a pipeline keeps
    one arena per backbone in flight and per schedule size).  Either way: run the frames in smaller launches
    (PosePipeline splits by itself)."""


ARENAS_PER_DEVICE = 4       # what a pipeline keeps: `depth` (2) backbones in flight x (the coalesced schedule + the batch-sized one for trailing groups)


def arena_budget(device):
    """Bytes ONE activation arena may take on `device`: SMAP_MAX_ARENA_BYTES, else 90 % of the device's TOTAL memory shared out over the
    arenas a pipeline keeps (SMAP_ARENAS_PER_DEVICE, default 4).  (Round 5 took 45 % of what happened to be FREE when the engine was built:
    the chunking of a pipeline -- hence its throughput -- then depended on the allocator's state and on which engine was built first.)"""
    env = os.environ.get("SMAP_MAX_ARENA_BYTES", "")
    if env:
        return int(float(env))
    _, total = torch.cuda.mem_get_info(device)
    return int(0.9 * total / max(1, int(os.environ.get("SMAP_ARENAS_PER_DEVICE", ARENAS_PER_DEVICE))))


class Graph:
    def __init__(self, sd, B, H, W, stage_num=3, chl=256, kpt_paf=43, paf=14, keep_ref=False, precision="f16",
                 flip_pair=None, build=True, scaled_hms=False):
        """build=False: an EMPTY schedule with all the book-keeping in place (single-op harnesses of tests / tools append to it with
        Graph.tensor / conv / conv_seg and then allocate() / emit()).
        flip_pair (43 ints: KEYPOINT.FLIP_ORDER + [15 + c for c in PAF.FLIP_CHANNEL]) switches the flip-TTA of
        test.py:55-70 on INSIDE the schedule: B input frames run as a 2B batch whose second half the stem reads mirrored,
        and the head sum merges the mirrored maps back (no flipped copy of the images, no separate merge pass); the
        depth heads, which the reference takes from the un-mirrored pass only, run on the first B frames."""
        assert precision in PRECISIONS, precision
        self.precision, self.x3 = precision, precision == "x3"
        self.keep_ref = keep_ref
        self.flip_pair = list(flip_pair) if flip_pair is not None else None
        # scaled_hms: the head sum stores hms / 255 (key points) and / 127 (PAFs), i.e. the maps as test.py:111-112 hands them to the
        # association (smap_op.scale_hms) -- what the pipelines ask for; SMAP.forward keeps the raw maps of smap.py:417-419
        self.scaled_hms = bool(scaled_hms)
        self.frames = B                       # frames of the input / output
        if self.flip_pair is not None:
            assert len(self.flip_pair) == kpt_paf
            if sorted(self.flip_pair) != list(range(kpt_paf)):      # the head sum indexes LDS with these values
                raise ValueError("flip_pair must be a permutation of the %d output channels (cfg FLIP_ORDER / PAF.FLIP_CHANNEL)" % kpt_paf)
            B = 2 * B                         # frames of every activation tensor
        assert not build or (H % 32 == 0 and W % 32 == 0), "input must be a multiple of 32 (5 stride-2 levels)"
        self.sd, self.B, self.H, self.W = sd, B, H, W
        self.w_pairs = use_w_pairs(B, H, W)                  # layout of the packed 32-half weight tiles (whole schedule)
        self.lanes = use_lanes(B, H, W)                      # head chains on forked streams (BackboneEngine switches the plan's lanes on)
        self.cur_lane = 0                                    # lane of the ops being appended (Graph.on_lane)
        self.ops, self.tensors = [], []
        self.scratch_tensors, self.kcount, self.kcount_tiles = [], None, 0      # split K: partial-tile scratch per op, one ticket region per schedule
        self.wchunks, self.woff = [], 0
        self.stage_num, self.chl, self.kpt_paf, self.paf = stage_num, chl, kpt_paf, paf
        self.flops = 0
        self.alg_bytes = 0                    # unfused per-launch traffic: input + output + weights + residual / skip adds / taps
        if build:
            self._build()

    # -- helpers
    def tensor(self, name, H, W, C, esize=2):
        t = Tensor(name, self.B, H, W, C, esize, 2 if (self.x3 and esize == 2) else 1)
        self.tensors.append(t)
        return t

    def deep_pipeline_tile(self, tile, M, cout_pad, K):
        """Tile 2 (64x64, two LDS stages of 64-half K tiles, 64 KiB) -> tile 7 (the same tile with FOUR stages: three K tiles in flight,
        128 KiB) for launches of at most 256 workgroups.  Such a launch leaves every CU to one workgroup anyway, so the second
        workgroup's worth of LDS buys prefetch depth instead: the K loops of these launches are bound by the latency of the ONE tile a
        two-stage pipeline keeps in flight (~1 us per K tile at batch 1).  Measured at batch 1 per shape (profiles/r5_v9_*): -6 .. -21 %
        on launches of <= 256 workgroups, +31 .. +41 % on launches of 416 (one workgroup per CU instead of two).  SMAP_DEEP_TILE=0: off."""
        if not self.x3 or tile != 2 or os.environ.get("SMAP_DEEP_TILE", "1") == "0":
            return tile
        bm, bn = TILES[2]
        wgs = -(-M // bm) * (cout_pad // bn) * self.split_k(2, M, cout_pad, K)
        return 7 if wgs <= 256 else 2

    def on_lane(self, lane):
        """Context manager: ops appended inside run on stream lane `lane` (0 when the schedule has no lanes)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self.cur_lane = self.cur_lane, (lane if self.lanes else 0)
            n0 = len(self.ops)
            try:
                yield
            finally:
                for op in self.ops[n0:]:
                    op.lane = self.cur_lane
                self.cur_lane = prev
        return cm()

    def split_k(self, tile, M, cout_pad, K):
        """K parts per output tile for a conv.hip launch of this shape (include/smap_hip.h smap_op.ksplit), or 1.  Small schedules only
        (batch 1: configs[1]): a launch with fewer output tiles than half the CUs whose K loop is long enough to share out -- the 32x52
        and 16x26 levels run 26-104 workgroups of 16-72 K tiles each there.  SMAP_SPLITK=0 switches it off, SMAP_SPLITK=<n> forces n
        parts wherever the tile allows."""
        env = os.environ.get("SMAP_SPLITK", "")
        if env == "0" or tile not in SPLITK_TILES:
            return 1
        bm, bn = TILES[tile]
        tiles = -(-M // bm) * (cout_pad // bn)
        n_k = K // tile_bk(tile, self.x3)
        if tiles >= 128 or n_k < 8:                      # the launch fills half the chip by itself, or has no K loop to share out
            return 1
        if env.startswith("t"):                          # "t384": aim at that many workgroups per launch, any K (A/B hook)
            s = min(8, int(env[1:]) // tiles, n_k // 4)
        elif env and env != "1":                         # "3": that many parts wherever a launch qualifies (tests)
            s = min(int(env), 16, n_k)
        else:
            # Measured at batch 1 (profiles/r5_v3_*): a part costs ~9 us (partial tile to memory and back, ticket, late epilogue), so
            # only long K loops gain: K = 4608 with 4 parts 57.6 -> 33.0 us, K = 2304 with 2 parts 30.2 -> 25.1, K = 2048 with 4
            # parts 27.9 -> 25.8; K = 1024 with 2 parts LOSES (17.0 -> 19.0), and aiming at 384 / 512 workgroups instead of 256 loses too
            if K < int(os.environ.get("SMAP_SPLITK_MINK", "2048")):      # (A/B hook; re-measured with the release / acquire ticket: R6.5)
                return 1
            s = min(4, 256 // tiles, n_k // 8)
        return s if s >= 2 else 1

    def _split_k_fields(self, name, tile, M, cout_pad, ks):
        """Scratch tensor + ticket slice of a split-K op -> (dict for Op.p, [scratch tensor])."""
        if ks <= 1:
            return {}, []
        bm, bn = TILES[tile]
        tiles = -(-M // bm) * (cout_pad // bn)
        part = Tensor(name + ".kpart", 1, 1, 1, tiles * ks * bm * bn, 4, 1)
        self.scratch_tensors.append(part)
        if self.kcount is None:
            self.kcount = Tensor("split_k.tickets", 1, 1, 1, 0, 4, 1)
        first_ticket = self.kcount_tiles
        self.kcount_tiles += tiles
        self.kcount.C = self.kcount_tiles
        return dict(ksplit=ks, kcount_first=first_ticket), [part]

    def _add_w(self, t):
        t = t.contiguous()
        raw = t.view(torch.uint8).reshape(-1) if t.dtype != torch.uint8 else t.reshape(-1)
        off = self.woff
        self.wchunks.append((off, raw))
        self.woff = _rup(off + raw.numel(), ALIGN)
        return off

    def conv(self, name, prefixes, x, ksize=1, stride=1, relu=True, res=None, add1=None, add2=None,
             in_c_off=0, cin=None, out_fp32=False, up=None, frames=None):
        """One conv launch; `prefixes` (list) are concatenated along Cout (shared input).  frames: run on the first
        `frames` frames of the batch only (flip-TTA: heads whose mirrored half nobody reads)."""
        ws, bs = zip(*[fold_conv_bn(self.sd, p) for p in prefixes])
        w, b = torch.cat(ws, 0), torch.cat(bs, 0)
        cout, cin_w = w.shape[0], w.shape[1]
        cin = cin or x.C
        assert cin_w == cin and w.shape[2] == ksize, (name, w.shape, cin, ksize)
        pad = ksize // 2
        Ho = (x.H + 2 * pad - ksize) // stride + 1
        Wo = (x.W + 2 * pad - ksize) // stride + 1
        nfr = self.B if frames is None else frames
        M = nfr * Ho * Wo
        key = f"{nfr},{x.H},{x.W},{cin},{cout},{ksize},{stride}"
        plain3 = ksize == 3 and stride == 1 and res is None and add1 is None and add2 is None and up is None
        legal = lambda t: tile_legal(t, cout=cout, plain3=plain3, up=up is not None, out_fp32=out_fp32,
                                     adds=add1 is not None or add2 is not None, x3=self.x3)
        if self.x3:
            # ops with the fused bilinear add have their own entries (the tap loads of the epilogue favour wider tiles)
            cands = pick_tile_x3(M, cout, [key + ",up", key] if up is not None else key)
            x3t = os.environ.get("SMAP_X3_TILE", "")         # A/B hook: force one split-precision tile where it fits
            if x3t and cout > 32 and not (cout <= 64 and TILES[int(x3t)][1] > 64):
                cands = [int(x3t)] + cands
            tile = next(t for t in cands if legal(t))
            tile = _tile_remap().get(tile, tile)             # A/B hook (SMAP_TILE_REMAP="2:7"): same tile shape, other pipeline depth
        else:
            tile = pick_tile(M, cout, key)
            tile = {**DEFAULT_REMAP, **_tile_remap()}.get(tile, tile)
            if not legal(tile):                                 # the table is keyed by shape only
                tile = pick_tile_heuristic(M, cout)
        halo = os.environ.get("SMAP_HALO3", "")     # A/B hook: "16" / "32" = pixel-tile width, optional ":64" / ":128" = BN
        if halo and plain3 and cout > 32:
            tw, _, hbn = halo.partition(":")
            hbn = int(hbn) if hbn else min(TILES[tile][1], 128 if cout > 64 else 64)
            tile = {(16, 64): 30, (16, 128): 31, (32, 64): 32, (32, 128): 33}[(int(tw), max(hbn, 64))]
            tile += 4 if os.environ.get("SMAP_HALO3_DEEP") else 0
        tile = self.deep_pipeline_tile(tile, M, _rup(cout, TILES[tile][1]), ksize * ksize * cin)
        bn = TILES[tile][1]
        cout_pad = _rup(cout, bn)
        K = ksize * ksize * cin
        acc_scale = 1.0
        if self.x3:                     # [cout_pad][K] hi | [cout_pad][K] lo of w * 2^s
            hi, lo, acc_scale = split_f16(w.permute(0, 2, 3, 1).reshape(cout, K))
            wk = torch.zeros((2, cout_pad, K), dtype=torch.float16)
            wk[0, :cout], wk[1, :cout] = hi, lo
        else:
            wk = torch.zeros((1, cout_pad, K), dtype=torch.float16)
            wk[0, :cout] = w.permute(0, 2, 3, 1).reshape(cout, K).to(torch.float16)
            if not torch.isfinite(wk).all():
                raise ValueError(f"{name}: folded weights exceed the fp16 range (max |w| = {float(w.abs().max()):.3g})")
        w_pairs = self.w_pairs
        wk = pack_conv_weights(wk, tile, self.x3, ksize, cin, pairs=w_pairs)   # one contiguous block per staged weight tile (pair)
        bk = torch.zeros((cout_pad,), dtype=torch.float32)
        bk[:cout] = b.to(torch.float32)
        out = self.tensor(name, Ho, Wo, _rup(cout, 8), 4 if out_fp32 else 2)
        fl = 2 * M * cout * K
        by = (nfr * x.H * x.W * cin * 2 * x.planes + nfr * Ho * Wo * out.C * out.esize * out.planes
              + wk.numel() * 2 + sum(t.nbytes * nfr // self.B for t in (res, add1, add2, up) if t is not None))
        self.flops += fl
        self.alg_bytes += by
        skf, scratch = self._split_k_fields(name, tile, M, cout_pad, self.split_k(tile, M, cout_pad, K))
        self.ops.append(Op(OP_CONV, out=out, inp=x, res=res, add1=add1, add2=add2, aux=[up] if up is not None else [], scratch=scratch, p=dict(
            flops=fl, alg_bytes=by, kinds="1x1" if ksize == 1 else "3x3", **skf,
            Cin=cin, in_c_off=in_c_off, Cout=cout, ksize=ksize, stride=stride, pad=pad, relu=int(relu),
            cout_pad=cout_pad, tile=tile, out_fp32=int(out_fp32), w_off=self._add_w(wk), bias_off=self._add_w(bk),
            acc_scale=acc_scale, frames=nfr, w_pairs=w_pairs,
            w_ref=w if self.keep_ref else None, b_ref=b if self.keep_ref else None)))
        return out

    def conv_relusum(self, name, pre1, x, pre2, x2, tile=None):
        """relu(conv1(x)) + relu(conv2(x2)) as ONE launch and ONE tensor (include/smap_hip.h smap_op.in2_mode = 1): the two inter-stage skips of an
        Upsample_unit, skip1 on the unit's input and skip2 on its output (smap.py:218-241), which the next stage only ever ADDS to its feature map
        (smap.py:142-153).  One write and one read of an in_planes-wide tensor less per level; each conv keeps its own power-of-two scale."""
        (w1, b1), (w2, b2) = fold_conv_bn(self.sd, pre1), fold_conv_bn(self.sd, pre2)
        cout, c1, c2 = w1.shape[0], w1.shape[1], w2.shape[1]
        assert w2.shape[0] == cout and w1.shape[2] == 1 == w2.shape[2] and c1 == x.C and c2 == x2.C and cout % 8 == 0 and (x.H, x.W) == (x2.H, x2.W)
        M, K = self.B * x.H * x.W, c1 + c2
        if tile is None:
            key = f"{self.B},{x.H},{x.W},{c1}+{c2}relusum,{cout},1,1"
            # measured in situ (profiles/r6_v14_ab_two_input_tiles.log): the 128 x 256 tile (both inputs' rows staged once per launch instead of once
            # per 128-channel N tile) 860-864 frames/s against 847 with the 128 x 128 tiles; the K-concatenated launches (conv_cat) do not care
            cands = (pick_tile_x3(M, cout, key) if self.x3 else []) + [54, 53, 50, 51]
            forced = os.environ.get("SMAP_RELUSUM_TILE", "")                     # A/B hook
            tile = int(forced) if forced else next(t for t in cands if t in RELUSUM_TILES)
        assert tile in RELUSUM_TILES, tile
        bn = TILES[tile][1]
        cout_pad = _rup(cout, bn)
        sc1 = sc2 = 1.0
        if self.x3:
            h1, l1, sc1 = split_f16(w1.reshape(cout, c1))
            h2, l2, sc2 = split_f16(w2.reshape(cout, c2))
            wk = torch.zeros((2, cout_pad, K), dtype=torch.float16)
            wk[0, :cout], wk[1, :cout] = torch.cat([h1, h2], 1), torch.cat([l1, l2], 1)
        else:
            wk = torch.zeros((1, cout_pad, K), dtype=torch.float16)
            wk[0, :cout] = torch.cat([w1.reshape(cout, c1), w2.reshape(cout, c2)], 1).to(torch.float16)
            if not torch.isfinite(wk).all():
                raise ValueError(f"{name}: folded weights exceed the fp16 range")
        wk = pack_conv_weights(wk, tile, self.x3, 1, K, pairs=self.w_pairs)
        bk1, bk2 = torch.zeros((cout_pad,), dtype=torch.float32), torch.zeros((cout_pad,), dtype=torch.float32)
        bk1[:cout], bk2[:cout] = b1.to(torch.float32), b2.to(torch.float32)
        out = self.tensor(name, x.H, x.W, cout)
        fl = 2 * M * cout * K
        by = x.nbytes + x2.nbytes + out.nbytes + wk.numel() * 2
        self.flops += fl
        self.alg_bytes += by
        keep = self.keep_ref
        self.ops.append(Op(OP_CONV, out=out, inp=x, aux2=x2, p=dict(
            flops=fl, alg_bytes=by, kinds="1x1",
            Cin=c1, in_c_off=0, Cout=cout, ksize=1, stride=1, pad=0, relu=0, cout_pad=cout_pad, tile=tile, out_fp32=0,
            w_off=self._add_w(wk), bias_off=self._add_w(bk1), acc_scale=sc1, frames=self.B, w_pairs=self.w_pairs,
            cat=dict(cin=c2, stride=1, relusum=True, acc_scale=sc2, bias_off=self._add_w(bk2), w_ref=w2 if keep else None, b_ref=b2 if keep else None),
            w_ref=w1 if keep else None, b_ref=b1 if keep else None)))
        return out

    def conv_tapdot(self, name, pre1, pre3, x, frames=None):
        """A 1x1 conv + ReLU (prefix pre1, 256 -> 256) whose ONLY consumer is a 3x3 conv with one output channel (prefix pre3) -- smap.py:227-229,
        res_rd_conv1 -> res_rd_conv2 -- with the per-pixel half of that 3x3 inside its epilogue (include/smap_hip.h smap_op.tap_n): the launch
        stores t[m][k] = <w3[k], y[m]>, nine fp32 numbers per pixel, instead of the 256-channel activation; Graph.tapsum finishes the conv.
        Returns (t tensor, bias of the 3x3)."""
        w1, b1 = fold_conv_bn(self.sd, pre1)
        w3, b3 = fold_conv_bn(self.sd, pre3)
        cout, cin = w1.shape[0], w1.shape[1]
        assert cout == 256 == w3.shape[1] and w3.shape[0] == 1 and w3.shape[2] == 3 and cin == x.C and w1.shape[2] == 1
        tile = 54
        nfr = self.B if frames is None else frames
        M = nfr * x.H * x.W
        acc_scale = 1.0
        if self.x3:
            hi, lo, acc_scale = split_f16(w1.reshape(cout, cin))
            wk = torch.stack([hi, lo])
        else:
            wk = w1.reshape(1, cout, cin).to(torch.float16)
        wk = pack_conv_weights(wk, tile, self.x3, 1, cin, pairs=self.w_pairs)
        # the 3x3's weights as B fragments of v_mfma_f32_16x16x32_f16 (include/smap_hip.h smap_op.tap_n): [K step][hi, lo][lane][8 halves],
        # lane l = tap l % 16 (9..15: zeros), channels 32 step + 8 (l / 16) .. +7; pre-scaled by a power of two like every split weight matrix
        tw = torch.zeros((16, cout), dtype=torch.float64)
        tw[:9] = w3[0].permute(1, 2, 0).reshape(9, cout)                                   # [kh*3+kw][channel]
        thi, tlo, tap_scale = split_f16(tw)
        lane = torch.arange(64)
        kidx = (torch.arange(cout // 32)[:, None, None] * 32 + (lane // 16)[None, :, None] * 8 + torch.arange(8)[None, None, :])     # [step][lane][8]
        tapw = torch.stack([thi[(lane % 16)[None, :, None], kidx], tlo[(lane % 16)[None, :, None], kidx]], 1).contiguous()              # [step][2][64][8]
        t = Tensor(name, self.B, x.H, x.W, 16, 4, 1)
        self.tensors.append(t)
        fl = 2 * M * cout * cin + 2 * M * 9 * cout
        by = nfr * x.H * x.W * cin * 2 * x.planes + nfr * x.H * x.W * 16 * 4 + wk.numel() * 2
        self.flops += fl
        self.alg_bytes += by
        keep = self.keep_ref
        self.ops.append(Op(OP_CONV, out=t, inp=x, p=dict(
            flops=fl, alg_bytes=by, kinds="1x1",
            Cin=cin, in_c_off=0, Cout=cout, ksize=1, stride=1, pad=0, relu=1, cout_pad=cout, tile=tile, out_fp32=1,
            w_off=self._add_w(wk), bias_off=self._add_w(b1.to(torch.float32)), acc_scale=acc_scale, frames=nfr, w_pairs=self.w_pairs,
            tap=dict(w_off=self._add_w(tapw), scale=tap_scale, w_ref=w3 if keep else None),
            w_ref=w1 if keep else None, b_ref=b1 if keep else None)))
        return t, b3

    def tapsum(self, t, b3, ext_off):
        """The stencil half of conv_tapdot's 3x3: out[b,0,y,x] = b3 + sum over taps of t[b, y+kh-1, x+kw-1][3 kh + kw] -> the fp32 NCHW map at
        ext_off of the output buffer (SMAP_OP_TAPSUM; stands where the head sum of that map stood)."""
        self.ops.append(Op(OP_TAPSUM, aux=[t], p=dict(Cout=1, ext_off=ext_off, bias_off=self._add_w(b3.to(torch.float32).reshape(1)),
                                                       b_ref=b3 if self.keep_ref else None)))

    def conv_cat(self, name, pre1, x, pre2, x2, stride2, relu=True, tile=None):
        """The last 1x1 of a Bottleneck (prefix pre1, on x) TOGETHER with the block's 1x1 shortcut conv (prefix pre2, on x2 sampled with spatial
        stride `stride2`) as ONE launch: out = act(W1 x + W2 x2 + b1 + b2) -- smap.py:60-77 adds the shortcut's output before the ReLU, no
        activation in between, so the two GEMMs share their accumulators when K is the concatenation of both inputs' channels
        (include/smap_hip.h smap_op.in2_*).  The shortcut tensor is never written or read back and its launch disappears."""
        (w1, b1), (w2, b2) = fold_conv_bn(self.sd, pre1), fold_conv_bn(self.sd, pre2)
        cout, c1, c2 = w1.shape[0], w1.shape[1], w2.shape[1]
        assert w2.shape[0] == cout and w1.shape[2] == 1 == w2.shape[2] and c1 == x.C and c2 == x2.C and cout % 8 == 0
        Ho, Wo = x.H, x.W
        assert (x2.H - 1) // stride2 + 1 == Ho and (x2.W - 1) // stride2 + 1 == Wo
        M, K = self.B * Ho * Wo, c1 + c2
        if tile is None:
            key = f"{self.B},{Ho},{Wo},{c1}+{c2}cat,{cout},1,1"
            cands = (pick_tile_x3(M, cout, key) if self.x3 else []) + [50, 51, 20]
            forced = os.environ.get("SMAP_CAT_TILE", "")                         # A/B hook
            tile = int(forced) if forced else next(t for t in cands if t in DUAL_TILES)
        assert tile in DUAL_TILES, tile
        bn = TILES[tile][1]
        cout_pad = _rup(cout, bn)
        w = torch.cat([w1.reshape(cout, c1), w2.reshape(cout, c2)], 1)           # [cout][K = (x channels | x2 channels)]
        b = b1 + b2
        acc_scale = 1.0
        if self.x3:
            hi, lo, acc_scale = split_f16(w)                                       # ONE power-of-two scale: the two matrices share the accumulators
            wk = torch.zeros((2, cout_pad, K), dtype=torch.float16)
            wk[0, :cout], wk[1, :cout] = hi, lo
        else:
            wk = torch.zeros((1, cout_pad, K), dtype=torch.float16)
            wk[0, :cout] = w.to(torch.float16)
            if not torch.isfinite(wk).all():
                raise ValueError(f"{name}: folded weights exceed the fp16 range")
        wk = pack_conv_weights(wk, tile, self.x3, 1, K, pairs=self.w_pairs)
        bk = torch.zeros((cout_pad,), dtype=torch.float32)
        bk[:cout] = b.to(torch.float32)
        out = self.tensor(name, Ho, Wo, cout)
        fl = 2 * M * cout * K
        by = x.nbytes + self.B * Ho * Wo * c2 * 2 * x2.planes + out.nbytes + wk.numel() * 2      # (the strided shortcut touches the sampled pixels only)
        self.flops += fl
        self.alg_bytes += by
        keep = self.keep_ref
        self.ops.append(Op(OP_CONV, out=out, inp=x, aux2=x2, p=dict(
            flops=fl, alg_bytes=by, kinds="1x1",
            Cin=c1, in_c_off=0, Cout=cout, ksize=1, stride=1, pad=0, relu=int(relu), cout_pad=cout_pad, tile=tile, out_fp32=0,
            w_off=self._add_w(wk), bias_off=self._add_w(bk), acc_scale=acc_scale, frames=self.B, w_pairs=self.w_pairs,
            cat=dict(cin=c2, stride=stride2, w_ref=w2 if keep else None, b_ref=b2 if keep else None),
            w_ref=w1 if keep else None, b_ref=b1 if keep else None)))
        return out

    def conv_seg(self, segs, x, up=None, tile=None):
        """Several 1x1 stride-1 convs that read the SAME input as ONE launch with one output tensor per conv (include/smap_hip.h
        smap_op.seg_*; Upsample_unit, smap.py:210-241: u_skip | skip1 on x; skip2 | cross_conv | the next unit's up_conv on
        `out`).  segs: [(tensor name, prefix, relu)], up to three; `up` (the fused bilinear add) belongs to the first.  The weight
        matrix is the concatenation of the convs' rows, each segment starting on a multiple of the tile's N extent and carrying
        its own power-of-two scale in split precision, so every conv computes exactly what its own launch would.  Returns the
        output tensors in order."""
        assert 2 <= len(segs) <= 3
        folded = [fold_conv_bn(self.sd, pre) for _, pre, _ in segs]
        cin = x.C
        couts = [w.shape[0] for w, _ in folded]
        assert all(w.shape[1] == cin and w.shape[2] == 1 for w, _ in folded) and all(c % 8 == 0 for c in couts)
        M = self.B * x.H * x.W
        # table keys: the merged shape "B,H,W,Cin,c0+c1[+c2],1,1" (tools/autotune_seg.py), then -- a launch nobody tuned -- the entry of its
        # widest conv alone (same input, same K: the nearest measured relative)
        key = f"{self.B},{x.H},{x.W},{cin},{'+'.join(map(str, couts))},1,1"
        wide = f"{self.B},{x.H},{x.W},{cin},{max(couts)},1,1"
        legal = lambda t: (tile_family(t) == "igemm" and (t in X3_TILES or not self.x3) and t not in (3, 8)
                           and (t not in REGEPI_TILES or (self.x3 and up is None)))
        if tile is not None:
            assert legal(tile), tile
        elif self.x3:
            keys = ([key + ",up"] if up is not None else []) + [key] + ([wide + ",up"] if up is not None else []) + [wide]
            cands = pick_tile_x3(M, sum(couts), keys)
            x3t = os.environ.get("SMAP_X3_TILE", "")         # A/B hook, as in Graph.conv
            if x3t:
                cands = [int(x3t)] + cands
            tile = _tile_remap().get(next(t for t in cands if legal(t)), None) or next(t for t in cands if legal(t))
        else:
            tile = pick_tile(M, sum(couts), key)
            if not legal(tile):
                tile = pick_tile_heuristic(M, sum(couts))
        if tile == 2:                                            # (the deep-pipeline variant of the 64x64 tile for small launches)
            tile = self.deep_pipeline_tile(2, M, sum(_rup(c, 64) for c in couts), cin)
        bn = TILES[tile][1]
        starts, n = [], 0
        for c in couts:
            starts.append(n)
            n = _rup(n + c, bn)
        cout_pad = n
        planes = 2 if self.x3 else 1
        wk = torch.zeros((planes, cout_pad, cin), dtype=torch.float16)
        bk = torch.zeros((cout_pad,), dtype=torch.float32)
        scales = []
        for (w, b), st, c in zip(folded, starts, couts):
            if self.x3:
                hi, lo, sc = split_f16(w.reshape(c, cin))
                wk[0, st:st + c], wk[1, st:st + c] = hi, lo
            else:
                sc = 1.0
                wk[0, st:st + c] = w.reshape(c, cin).to(torch.float16)
            scales.append(sc)
            bk[st:st + c] = b.to(torch.float32)
        if not torch.isfinite(wk).all():
            raise ValueError(f"{segs[0][0]}: folded weights exceed the fp16 range")
        wk = pack_conv_weights(wk, tile, self.x3, 1, cin, pairs=self.w_pairs)
        outs = [self.tensor(nm, x.H, x.W, c) for (nm, _, _), c in zip(segs, couts)]
        fl = 2 * M * sum(couts) * cin
        by = x.nbytes + sum(t.nbytes for t in outs) + wk.numel() * 2 + (up.nbytes if up is not None else 0)
        self.flops += fl
        self.alg_bytes += by
        keep = self.keep_ref
        skf, scratch = self._split_k_fields(segs[0][0], tile, M, cout_pad, self.split_k(tile, M, cout_pad, cin))
        self.ops.append(Op(OP_CONV, out=outs[0], inp=x, aux=[up] if up is not None else [], outs=outs[1:], scratch=scratch, p=dict(
            flops=fl, alg_bytes=by, kinds="1x1", **skf,
            Cin=cin, in_c_off=0, Cout=couts[0], ksize=1, stride=1, pad=0, relu=int(segs[0][2]), cout_pad=cout_pad, tile=tile, out_fp32=0,
            w_off=self._add_w(wk), bias_off=self._add_w(bk), acc_scale=scales[0], frames=self.B, w_pairs=self.w_pairs,
            segs=[dict(n0=st, cout=c, relu=int(r), acc_scale=sc, w_ref=w if keep else None, b_ref=b if keep else None)
                  for st, c, (_, _, r), sc, (w, b) in list(zip(starts, couts, segs, scales, folded))[1:]],
            w_ref=folded[0][0] if keep else None, b_ref=folded[0][1] if keep else None)))
        return outs

    def conv_tail(self, name, pre3, pre1, x, tile, res=None, add1=None, add2=None):
        """A Bottleneck's 3x3 stride-1 conv (prefix pre3, bias + ReLU) and the 1x1 behind it (prefix pre1, + res, ReLU, + add1,
        + add2) as ONE launch (csrc/convf.hip, tile ids 80..89): the 3x3's output never leaves the CU."""
        w3, b3 = fold_conv_bn(self.sd, pre3)
        w1, b1 = fold_conv_bn(self.sd, pre1)
        P, cin = w3.shape[0], w3.shape[1]
        cout = w1.shape[0]
        bn2 = TAIL_BN[tile]
        assert TILES[tile][1] == P == w1.shape[1] and cin == x.C and w3.shape[2] == 3 and w1.shape[2] == 1 and cout % 8 == 0
        cout_pad = _rup(cout, bn2)
        M = self.B * x.H * x.W
        K3 = 9 * cin
        sc3 = sc1 = 1.0
        if self.x3:
            hi, lo, sc3 = split_f16(w3.permute(0, 2, 3, 1).reshape(P, K3))
            wk3 = torch.stack([hi, lo])
            hi, lo, sc1 = split_f16(w1.reshape(cout, P))
            wk1 = torch.zeros((2, cout_pad, P), dtype=torch.float16)
            wk1[0, :cout], wk1[1, :cout] = hi, lo
        else:
            wk3 = w3.permute(0, 2, 3, 1).reshape(1, P, K3).to(torch.float16)
            wk1 = torch.zeros((1, cout_pad, P), dtype=torch.float16)
            wk1[0, :cout] = w1.reshape(cout, P).to(torch.float16)
            if not (torch.isfinite(wk3).all() and torch.isfinite(wk1).all()):
                raise ValueError(f"{name}: folded weights exceed the fp16 range")
        wk3 = pack_halo_rows(wk3, P, 9, cin, self.x3)
        wk1 = pack_halo_rows(wk1, bn2, 1, P, self.x3)
        bk1 = torch.zeros((cout_pad,), dtype=torch.float32)
        bk1[:cout] = b1.to(torch.float32)
        out = self.tensor(name, x.H, x.W, cout)
        fl = 2 * M * (P * K3 + cout * P)
        by = (x.nbytes + out.nbytes + (wk3.numel() + wk1.numel()) * 2
              + sum(t.nbytes for t in (res, add1, add2) if t is not None))
        self.flops += fl
        self.alg_bytes += by
        self.ops.append(Op(OP_CONV, out=out, inp=x, res=res, add1=add1, add2=add2, p=dict(
            flops=fl, alg_bytes=by, kinds="3x3+1x1",
            Cin=cin, in_c_off=0, Cout=P, ksize=3, stride=1, pad=1, relu=1, cout_pad=P, tile=tile, out_fp32=0,
            w_off=self._add_w(wk3), bias_off=self._add_w(b3.to(torch.float32)), acc_scale=sc3, frames=self.B, w_pairs=0,
            tail=dict(cout=cout, cout_pad=cout_pad, w_off=self._add_w(wk1), bias_off=self._add_w(bk1), acc_scale=sc1,
                      w_ref=w1 if self.keep_ref else None, b_ref=b1 if self.keep_ref else None),
            w_ref=w3 if self.keep_ref else None, b_ref=b3 if self.keep_ref else None)))
        return out

    def conv_block_first(self, name, pre, x, tile):
        """The FIRST Bottleneck of layer1 (smap.py:48-77 with the 1x1 shortcut conv of :124-129; 64 input channels, stride 1) as ONE
        launch (csrc/convb.hip, tile ids 92, 93): relu(c3(c2(c1(x))) + downsample(x)); x is read once."""
        assert self.x3
        w1, b1 = fold_conv_bn(self.sd, pre + ".conv_bn_relu1")
        w3, b3 = fold_conv_bn(self.sd, pre + ".conv_bn_relu2")
        wt, bt = fold_conv_bn(self.sd, pre + ".conv_bn_relu3")
        wd, bd = fold_conv_bn(self.sd, pre + ".downsample")
        P, Cin = w1.shape[0], w1.shape[1]
        C = wt.shape[0]
        assert P == 64 == Cin == x.C and C == 256 and tuple(wd.shape[:2]) == (C, Cin) and wd.shape[2] == 1 and w3.shape[2] == 3
        M = self.B * x.H * x.W
        hi, lo, sc1 = split_f16(w1.reshape(P, Cin))
        wk1 = pack_halo_rows(torch.stack([hi, lo]), P, 1, Cin, True)                  # [1][2 chunks][1][64 rows][128 B]
        hi, lo, sc3 = split_f16(w3.permute(0, 2, 3, 1).reshape(P, 9 * P))
        wk3 = pack_halo_rows(torch.stack([hi, lo]), P, 9, P, True)
        hi, lo, sct = split_f16(wt.reshape(C, P))
        wkt = pack_halo_rows(torch.stack([hi, lo]), 64, 1, P, True)
        hi, lo, scd = split_f16(wd.reshape(C, Cin))
        wkd = pack_halo_rows(torch.stack([hi, lo]), 64, 1, Cin, True)                 # [4 n chunks][2 k chunks][1][64 rows][128 B]
        out = self.tensor(name, x.H, x.W, C)
        fl = 2 * M * (P * Cin + P * 9 * P + C * P + C * Cin)
        by = x.nbytes + out.nbytes + (wk1.numel() + wk3.numel() + wkt.numel() + wkd.numel()) * 2
        self.flops += fl
        self.alg_bytes += by
        keep = self.keep_ref
        self.ops.append(Op(OP_CONV, out=out, inp=x, p=dict(
            flops=fl, alg_bytes=by, kinds="block",
            Cin=P, in_c_off=0, Cout=P, ksize=3, stride=1, pad=1, relu=1, cout_pad=P, tile=tile, out_fp32=0,
            w_off=self._add_w(wk3), bias_off=self._add_w(b3.to(torch.float32)), acc_scale=sc3, frames=self.B, w_pairs=0,
            head=dict(cin=Cin, w_off=self._add_w(wk1), bias_off=self._add_w(b1.to(torch.float32)), acc_scale=sc1,
                      w_ref=w1 if keep else None, b_ref=b1 if keep else None),
            tail=dict(cout=C, cout_pad=C, w_off=self._add_w(wkt), bias_off=self._add_w((bt + bd).to(torch.float32)), acc_scale=sct,
                      w_ref=wt if keep else None, b_ref=bt if keep else None),
            short=dict(w_off=self._add_w(wkd), acc_scale=scd, w_ref=wd if keep else None, b_ref=bd if keep else None),
            w_ref=w3 if keep else None, b_ref=b3 if keep else None)))
        return out

    def conv_block(self, name, pre, x, tile, add1=None, add2=None):
        """A whole stride-1 identity Bottleneck (smap.py:48-77: conv_bn_relu1 1x1 -> conv_bn_relu2 3x3 -> conv_bn_relu3 1x1, + x,
        ReLU, + add1, + add2) as ONE launch (csrc/convb.hip, tile ids 90..99, split precision): x is read once, the two
        intermediates and the residual never leave the CU."""
        assert self.x3, "the whole-block kernel has a split-precision instance only"
        w1, b1 = fold_conv_bn(self.sd, pre + ".conv_bn_relu1")
        w3, b3 = fold_conv_bn(self.sd, pre + ".conv_bn_relu2")
        wt, bt = fold_conv_bn(self.sd, pre + ".conv_bn_relu3")
        P, C = w1.shape[0], w1.shape[1]
        bn2 = TAIL_BN[tile]
        assert P == TILES[tile][1] and C == 4 * P == x.C == wt.shape[0] and w3.shape[:2] == (P, P) and w3.shape[2] == 3 and wt.shape[1] == P
        assert (P == 128) == (tile == 94) and P in (64, 128)
        M = self.B * x.H * x.W
        hi, lo, sc1 = split_f16(w1.reshape(P, C))
        if P == 64:
            wk1 = pack_rows16(torch.stack([hi, lo]))                                  # [C/16 stages][P rows][64 B]
        else:
            wk1 = pack_halo_rows(torch.stack([hi, lo]), P, 1, C, True)                # csrc/convc.hip: [1][C/32 chunks][1][P rows][128 B]
        hi, lo, sc3 = split_f16(w3.permute(0, 2, 3, 1).reshape(P, 9 * P))
        wk3 = pack_halo_rows(torch.stack([hi, lo]), P, 9, P, True)                    # [1][P/32][9 taps][P rows][128 B]
        hi, lo, sct = split_f16(wt.reshape(C, P))
        wkt = pack_halo_rows(torch.stack([hi, lo]), bn2, 1, P, True)                  # [C/64][P/32][1][64 rows][128 B]
        out = self.tensor(name, x.H, x.W, C)
        fl = 2 * M * (P * C + P * 9 * P + C * P)
        by = (x.nbytes + out.nbytes + (wk1.numel() + wk3.numel() + wkt.numel()) * 2
              + sum(t.nbytes for t in (add1, add2) if t is not None))
        self.flops += fl
        self.alg_bytes += by
        keep = self.keep_ref
        self.ops.append(Op(OP_CONV, out=out, inp=x, res=x, add1=add1, add2=add2, p=dict(
            flops=fl, alg_bytes=by, kinds="block",
            Cin=P, in_c_off=0, Cout=P, ksize=3, stride=1, pad=1, relu=1, cout_pad=P, tile=tile, out_fp32=0,
            w_off=self._add_w(wk3), bias_off=self._add_w(b3.to(torch.float32)), acc_scale=sc3, frames=self.B, w_pairs=0,
            head=dict(cin=C, w_off=self._add_w(wk1), bias_off=self._add_w(b1.to(torch.float32)), acc_scale=sc1,
                      w_ref=w1 if keep else None, b_ref=b1 if keep else None),
            tail=dict(cout=C, cout_pad=C, w_off=self._add_w(wkt), bias_off=self._add_w(bt.to(torch.float32)), acc_scale=sct,
                      w_ref=wt if keep else None, b_ref=bt if keep else None),
            w_ref=w3 if keep else None, b_ref=b3 if keep else None)))
        return out

    def block_tile(self, planes, stride, has_ds):
        """Tile id of the whole-block launch for this Bottleneck, or None.  Identity blocks (no shortcut conv) of stride 1 in
        split precision only; SMAP_BLOCK="64:91" chooses per width (A/B hook), default BLOCK_DEFAULT -- in small schedules (<= 2 frames of
        512x832) without layer2's entry (at batch 1 the 64x104 level has 52 tiles of the eight-wave kernel for 256 CUs, 55-57 us per block
        where the three launches take ~42: profiles/r5_v2_x3_layers_batch1.txt) and with layer1's on 4 x 16 pixel tiles (twice the
        workgroups: profiles/r5_v11_ab_b1_whole_block_tiles.log)."""
        if stride != 1 or has_ds or not self.x3:
            return None
        spec = os.environ.get("SMAP_BLOCK")
        table = BLOCK_DEFAULT if spec is None else {int(k): int(v) for k, v in (kv.split(":") for kv in spec.split(",") if ":" in kv)}
        small = self.B * self.H * self.W <= 2 * 512 * 832
        if spec is None and planes == 128 and small:
            return None
        if spec is None and planes == 64 and small:
            return 90                    # 4 x 16 pixel tiles: 416 workgroups of half the work at batch 1 (2.91 -> 2.84 ms per frame)
        return table.get(planes)

    # -- the network (smap.py:313-353 structure, :403-419 data flow)
    def _build(self):
        B, H, W, sd = self.B, self.H, self.W, self.sd
        # ResNet_top (smap.py:80-92)
        w, b = fold_conv_bn(sd, "top.conv")
        stem_scale = 1.0
        if self.x3:
            hi, lo, stem_scale = split_f16(w.permute(0, 2, 1, 3).reshape(64, 21, 7))
            wk = torch.zeros((2, 64, 22, 8), dtype=torch.float16)
            wk[0, :, :21, :7], wk[1, :, :21, :7] = hi, lo
            wk = wk.reshape(2, 64, 176)
        else:
            wk = torch.zeros((64, 22, 8), dtype=torch.float16)      # K = (kh, c, kw 7->8) + one zero granule
            wk[:, :21, :7] = w.permute(0, 2, 1, 3).reshape(64, 21, 7).to(torch.float16)
            wk = wk.reshape(64, 176)
        H2, W2 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        H4, W4 = (H2 + 2 - 3) // 2 + 1, (W2 + 2 - 3) // 2 + 1
        self.flops += 2 * B * H2 * W2 * 64 * 147
        stem_p = dict(w_off=self._add_w(wk), bias_off=self._add_w(b.to(torch.float32)), w_ref=w, b_ref=b, acc_scale=stem_scale)
        if not os.environ.get("SMAP_STEMPOOL"):         # default: conv and max-pool as two kernels (the fused kernel below is
            t = self.tensor("top.conv", H2, W2, 64)     # bit-identical but no faster inside the two-batch pipeline)
            self.ops.append(Op(OP_STEM, out=t, p=stem_p))
            x = self.tensor("top.pool", H4, W4, 64)
            self.ops.append(Op(OP_MAXPOOL, out=x, inp=t))
        else:                                           # SMAP_STEMPOOL=1: ResNet_top in one kernel, only the pooled tensor is written
            x = self.tensor("top.pool", H4, W4, 64)
            self.ops.append(Op(OP_STEMPOOL, out=x, p=stem_p))
        self.out_h, self.out_w = H4, W4
        skip1 = skip2 = None
        for s in range(self.stage_num):
            last = s == self.stage_num - 1
            x, skip1, skip2 = self._stage(s, x, skip1, skip2, gen_skip=not last, heads=last)

    def _bottleneck(self, pre, x, planes, stride, has_ds, add1=None, add2=None):
        # Bottleneck (smap.py:48-77): stride on the 3x3, shortcut 1x1 stride-s when shape changes
        blk = self.block_tile(planes, stride, has_ds)
        if blk is not None:         # c1 + c2 + c3 + residual in one launch (csrc/convb.hip)
            return self.conv_block(pre + ".c3", pre, x, blk, add1=add1, add2=add2)
        spec = os.environ.get("SMAP_BLOCK_FIRST")
        first = (BLOCK_FIRST_DEFAULT if spec is None else {int(k): int(v) for k, v in (kv.split(":") for kv in spec.split(",") if ":" in kv)}).get(planes)
        if spec is None and first == 93 and self.B * self.H * self.W <= 2 * 512 * 832:
            first = 92                   # small schedules: the 4 x 16 tiles here too
        if first is not None and self.x3 and has_ds and stride == 1 and x.C == 64 and planes == 64 and add1 is None and add2 is None:
            return self.conv_block_first(pre + ".c3", pre, x, first)
        # A block with a shortcut conv (the first of layer2 / 3 / 4; layer1's runs as a whole-block launch above): its last 1x1 and the shortcut
        # as ONE GEMM over K = (planes | in_planes) -- the shortcut tensor is never stored (conv_cat).  SMAP_CAT=0: two launches, as round 5 ran.
        # (Not for arenas beyond one 4 GiB window: both inputs are addressed from one base.)
        # (Not in small schedules -- <= 2 frames of 512x832 --: their launches live on split K and the deep-pipeline 64 x 64 tile, which the
        #  two-input instances do not have: batch 1 measured 3.4 ms per frame with them against 2.9 without, profiles/r6_final_*.)
        cat_env = os.environ.get("SMAP_CAT", "")
        cat = (has_ds and add1 is None and add2 is None and cat_env != "0" and self.tail_tile(planes, stride) is None
               and self.B * self.H * self.W <= 20 * 512 * 832 and (cat_env == "1" or self.B * self.H * self.W > 2 * 512 * 832))
        idn = x if (not has_ds or cat) else self.conv(pre + ".downsample", [pre + ".downsample"], x, 1, stride, relu=False)
        y = self.conv(pre + ".c1", [pre + ".conv_bn_relu1"], x, 1, 1, relu=True)
        if cat:
            y = self.conv(pre + ".c2", [pre + ".conv_bn_relu2"], y, 3, stride, relu=True)
            return self.conv_cat(pre + ".c3", pre + ".conv_bn_relu3", y, pre + ".downsample", x, stride, relu=True)
        tail = self.tail_tile(planes, stride)
        if tail is not None:        # c2 + c3 in one launch (csrc/convf.hip)
            return self.conv_tail(pre + ".c3", pre + ".conv_bn_relu2", pre + ".conv_bn_relu3", y, tail, res=idn, add1=add1, add2=add2)
        y = self.conv(pre + ".c2", [pre + ".conv_bn_relu2"], y, 3, stride, relu=True)
        return self.conv(pre + ".c3", [pre + ".conv_bn_relu3"], y, 1, 1, relu=True, res=idn, add1=add1, add2=add2)

    def tail_tile(self, planes, stride):
        """Tile id of the fused 3x3 + 1x1 launch for a Bottleneck of this width, or None = two launches.
        SMAP_TAIL="64:80,128:82" chooses per width (A/B hook); default: see TAIL_DEFAULT."""
        if stride != 1:
            return None
        spec = os.environ.get("SMAP_TAIL")
        table = TAIL_DEFAULT if spec is None else {int(k): int(v) for k, v in (kv.split(":") for kv in spec.split(",") if ":" in kv)}
        return table.get(planes)

    def _stage(self, s, x, skip1, skip2, gen_skip, heads):
        pre = f"stage{s}."
        feats = []
        inpl = 64
        for li, (planes, nblk) in enumerate(zip(PLANES, LAYERS)):
            stride = 1 if li == 0 else 2
            for j in range(nblk):
                lastb = j == nblk - 1
                a1 = skip1[li] if (skip1 is not None and lastb) else None   # smap.py:142-153
                a2 = skip2[li] if (skip2 is not None and lastb) else None
                has_ds = j == 0 and (stride != 1 or inpl != planes * 4)
                x = self._bottleneck(f"{pre}downsample.layer{li + 1}.{j}", x, planes, stride if j == 0 else 1,
                                     has_ds, a1, a2)
                inpl = planes * 4
            feats.append(x)
        x1, x2, x3, x4 = feats
        # Upsample_module (smap.py:244-286): up1 on x4 ... up4 on x1
        out, s1, s2, cross = None, [None] * 4, [None] * 4, None
        head_t = {}
        # Shared-input 1x1s of an Upsample_unit as ONE launch with one output per conv (conv_seg).  Measured (profiles/r5_v1_*): the
        # launches on `out` (skip2 | cross_conv | res_conv1 | the next unit's up_conv) are 3-33 % faster merged than one by one at every
        # level; u_skip | skip1 on x is faster merged where u_skip has no fused bilinear add (up1: 228 vs 244 us at 16 frames) and SLOWER
        # where it has one (128x208: 712 vs 610 us; 32x52: 277 vs 256): one tile shape must then serve the bilinear epilogue (which wants
        # the 256-wide tile) and the plain skip1 (which wants the eight-wave 128x128 one).  SMAP_MERGE_1X1: "1" (default) = the merges
        # that pay, "2" = all of them, "0" = one launch per conv.
        merge_mode = os.environ.get("SMAP_MERGE_1X1", "1")
        merge = merge_mode != "0"
        tl = None                                                 # up_conv@low of the unit at hand (a segment of the previous unit's launch on `out`)
        for ind, xin in enumerate((x4, x3, x2, x1)):
            u = f"{pre}upsample.up{ind + 1}"
            un = f"{pre}upsample.up{ind + 2}"                     # the next unit: its up_conv reads this unit's `out` (commuted with the upsample)
            # The inter-stage skips as ONE tensor per level: skip1(x) + skip2(out) in one launch (conv_relusum); the next stage adds that one tensor.
            # SMAP_SKIPSUM=0: round 5's two tensors (skip1 beside u_skip, skip2 as a segment of the launch on `out`).  Not beyond one 4 GiB window.
            ss_env = os.environ.get("SMAP_SKIPSUM", "")
            skipsum = (gen_skip and merge and ss_env != "0" and self.B * self.H * self.W <= 20 * 512 * 832
                       and (ss_env == "1" or self.B * self.H * self.W > 2 * 512 * 832))          # (small schedules: as conv_cat, see _bottleneck)
            if merge:
                # launch 1, on x:   out = relu(u_skip(x) [+ bilinear(up_conv@low)])  |  skip1 = relu(skip1(x))
                if gen_skip and not skipsum and (tl is None or merge_mode == "2"):
                    out, s1[3 - ind] = self.conv_seg([(u + ".out", u + ".u_skip", True), (u + ".skip1", u + ".skip1", True)], xin, up=tl)
                else:
                    out = self.conv(u + ".out", [u + ".u_skip"], xin, relu=True, up=tl)
                    if gen_skip and not skipsum:
                        s1[3 - ind] = self.conv(u + ".skip1", [u + ".skip1"], xin, relu=True)
                if skipsum:
                    s1[3 - ind] = self.conv_relusum(u + ".skipsum", u + ".skip1", xin, u + ".skip2", out)
                # launch 2, on out: skip2 | cross_conv (stages with skips), res_conv1 (last stage), the next unit's up_conv@low
                sg = []
                if gen_skip:
                    if not skipsum:
                        sg.append((u + ".skip2", u + ".skip2", True))
                    if ind == 3:
                        sg.append((u + ".cross_conv", u + ".cross_conv", True))
                if heads and 1 <= ind < 3:
                    sg.append((u + ".res1", u + ".res_conv1", True))
                if ind < 3:
                    sg.append((un + ".up_conv@low", un + ".up_conv", False))
                got = {}
                if len(sg) >= 2:
                    got = dict(zip([n_ for n_, _, _ in sg], self.conv_seg(sg, out)))
                elif len(sg) == 1:
                    got = {sg[0][0]: self.conv(sg[0][0], [sg[0][1]], out, relu=sg[0][2])}
                tl = got.get(un + ".up_conv@low")
                if gen_skip:
                    s2[3 - ind] = None if skipsum else got[u + ".skip2"]
                    if ind == 3:
                        cross = got[u + ".cross_conv"]
                if heads:
                    if ind == 3:
                        # The root-depth head (res_rd_conv1 -> res_rd_conv2, a 3x3 with ONE output channel) as a 1x1 launch whose epilogue
                        # keeps nine dot products per pixel instead of the 256-channel activation, + a nine-term stencil (conv_tapdot / tapsum):
                        # 0.87 GB per 16 frames and the N = 1 MFMA launch (8.9 TFLOP/s) gone.  SMAP_TAPHEAD=0: round 5's three-way 1x1 + 3x3.
                        tap_env = os.environ.get("SMAP_TAPHEAD", "")
                        tap = self.chl == 256 and tap_env != "0" and (tap_env == "1" or self.B * self.H * self.W > 2 * 512 * 832)
                        m = self.conv(u + ".heads1x1", [u + ".res_conv1", u + ".res_d_conv1"] + ([] if tap else [u + ".res_rd_conv1"]), out, relu=True)
                        c = self.chl
                        head_t["res4"] = self.conv(u + ".res", [u + ".res_conv2"], m, 3, relu=False, in_c_off=0, cin=c, out_fp32=True)
                        with self.on_lane(1):     # the three 3x3 heads read disjoint channel slices of m: side by side
                            head_t["res_d"] = self.conv(u + ".res_d", [u + ".res_d_conv2"], m, 3, relu=False, in_c_off=c, cin=c, out_fp32=True,
                                                        frames=self.frames)
                        with self.on_lane(2):
                            if tap:
                                head_t["res_rd_tap"] = self.conv_tapdot(u + ".res_rd_t", u + ".res_rd_conv1", u + ".res_rd_conv2", out, frames=self.frames)
                            else:
                                head_t["res_rd"] = self.conv(u + ".res_rd", [u + ".res_rd_conv2"], m, 3, relu=False, in_c_off=2 * c, cin=c,
                                                             out_fp32=True, frames=self.frames)
                    elif ind >= 1:
                        with self.on_lane(1):     # the 3x3 head of this unit runs beside the next unit's launches
                            head_t[f"res{ind + 1}"] = self.conv(u + ".res", [u + ".res_conv2"], got[u + ".res1"], 3, relu=False, out_fp32=True)
                continue
            if ind == 0:
                out = self.conv(u + ".out", [u + ".u_skip"], xin, relu=True)
            else:
                tl = self.conv(u + ".up_conv@low", [u + ".up_conv"], out, relu=False)   # commuted with the upsample
                if os.environ.get("SMAP_NO_UPADD_FUSION") and not self.x3:
                    a = self.conv(u + ".u_skip", [u + ".u_skip"], xin, relu=False)
                    o = self.tensor(u + ".out", a.H, a.W, a.C)
                    self.ops.append(Op(OP_UPADD, out=o, inp=a, aux=[tl], p=dict(relu=1)))
                    out = o
                else:       # relu(u_skip(x) + bilinear(tl)) in the u_skip conv's epilogue
                    out = self.conv(u + ".out", [u + ".u_skip"], xin, relu=True, up=tl)
            if gen_skip:
                lvl = 3 - ind                      # skip lists are fine -> coarse (smap.py:283-284)
                s1[lvl] = self.conv(u + ".skip1", [u + ".skip1"], xin, relu=True)
                s2[lvl] = self.conv(u + ".skip2", [u + ".skip2"], out, relu=True)
                if ind == 3:
                    cross = self.conv(u + ".cross_conv", [u + ".cross_conv"], out, relu=True)
            if heads:
                if ind == 3:
                    m = self.conv(u + ".heads1x1", [u + ".res_conv1", u + ".res_d_conv1", u + ".res_rd_conv1"],
                                  out, relu=True)
                    c = self.chl
                    head_t["res4"] = self.conv(u + ".res", [u + ".res_conv2"], m, 3, relu=False, in_c_off=0,
                                               cin=c, out_fp32=True)
                    head_t["res_d"] = self.conv(u + ".res_d", [u + ".res_d_conv2"], m, 3, relu=False, in_c_off=c,
                                                cin=c, out_fp32=True, frames=self.frames)
                    head_t["res_rd"] = self.conv(u + ".res_rd", [u + ".res_rd_conv2"], m, 3, relu=False,
                                                 in_c_off=2 * c, cin=c, out_fp32=True, frames=self.frames)
                elif ind >= 1:
                    m = self.conv(u + ".res1", [u + ".res_conv1"], out, relu=True)
                    head_t[f"res{ind + 1}"] = self.conv(u + ".res", [u + ".res_conv2"], m, 3, relu=False,
                                                        out_fp32=True)
        if heads:
            B, h, w = self.frames, self.out_h, self.out_w
            n_hms, n_d = self.kpt_paf, self.paf
            flip_p = {}
            if self.flip_pair is not None:         # pair table of the in-schedule flip-TTA merge lives in the weight blob
                flip_p = dict(flip_from=self.frames, n_kpt=n_hms - 2 * n_d,
                              w_off=self._add_w(torch.tensor(self.flip_pair, dtype=torch.int32)))
            self.out_layout = dict(hms=(0, n_hms), det_d=(B * n_hms * h * w * 4, n_d),
                                   root_d=(B * (n_hms + n_d) * h * w * 4, 1))
            self.out_bytes = B * (n_hms + n_d + 1) * h * w * 4
            self.status_off = self.out_bytes                 # int32 status words behind the maps (include/smap_hip.h)
            self.status_words = (B + 30) // 31               # SMAP_STATUS_WORDS: frame f = word f // 31, bit 1 + f % 31
            # outputs_2d = res4 + res3 + res2 (smap.py:417)
            self.ops.append(Op(OP_HEADSUM, aux=[head_t["res4"], head_t["res3"], head_t["res2"]],
                               p={**dict(Cout=n_hms, ext_off=self.out_layout["hms"][0], n_kpt=n_hms - 2 * n_d, scale_hms=int(self.scaled_hms)),
                                  **flip_p}))
            with self.on_lane(1):
                self.ops.append(Op(OP_HEADSUM, aux=[head_t["res_d"]], p=dict(Cout=n_d, ext_off=self.out_layout["det_d"][0])))
            with self.on_lane(2):
                if "res_rd_tap" in head_t:
                    self.tapsum(*head_t["res_rd_tap"], self.out_layout["root_d"][0])
                else:
                    self.ops.append(Op(OP_HEADSUM, aux=[head_t["res_rd"]], p=dict(Cout=1, ext_off=self.out_layout["root_d"][0])))
        return cross, (s1 if gen_skip else None), (s2 if gen_skip else None)

    # -- arena: liveness-based first-fit allocation
    def allocate(self, reuse=True):
        for i, op in enumerate(self.ops):
            for t in [op.inp, op.res, op.add1, op.add2, op.aux2] + list(op.aux):
                if t is not None:
                    t.last = i
            for t in ([op.out] if op.out is not None else []) + list(op.outs):
                t.first = i
                t.last = max(t.last, i)
            for t in op.scratch:
                t.first = t.last = i
        # Lanes: an op on a side lane may start as soon as its producers are done and end as late as the schedule does.  What it WRITES is
        # therefore live from the op after its last producer, what it READS stays live to the end of the schedule (smap_op.lane's contract).
        producer = {}
        for i, op in enumerate(self.ops):
            for t in ([op.out] if op.out is not None else []) + list(op.outs):
                producer[id(t)] = i
        n_last = len(self.ops) - 1
        for i, op in enumerate(self.ops):
            if not op.lane:
                continue
            reads = [t for t in [op.inp, op.res, op.add1, op.add2, op.aux2] + list(op.aux) if t is not None]
            start = 1 + max([producer.get(id(t), -1) for t in reads] + [-1])
            for t in ([op.out] if op.out is not None else []) + list(op.outs) + list(op.scratch):
                t.first = min(t.first, start)
            for t in reads + list(op.scratch):
                t.last = n_last
        if self.kcount is not None:                                # the tickets of every split-K op: one region, alive for the whole schedule
            self.kcount.first, self.kcount.last = 0, len(self.ops) - 1
        every = self.tensors + self.scratch_tensors + ([self.kcount] if self.kcount is not None else [])
        # arena[k * WINDOW : k * WINDOW + ZERO_PAGE] are the conv kernels' zero pages (csrc/plan.hip): never allocated, so no
        # tensor crosses a window boundary and every conv input is within 32 bits of its window's base
        free, top = [], ZERO_PAGE    # free: list of (off, size)
        for t in every:
            if _rup(t.nbytes, ALIGN) > WINDOW - ZERO_PAGE:
                raise ArenaTooLarge(f"{t.name}: {t.nbytes / 2 ** 30:.2f} GiB ({self.B} frames, precision {self.precision}) does not fit a "
                                    "4 GiB addressing window of the conv kernels -- use a smaller batch (PosePipeline splits by itself)")
        by_first = {}
        for t in every:
            by_first.setdefault(t.first, []).append(t)
        expiring = {}
        for t in every:
            expiring.setdefault(t.last, []).append(t)
        for i in range(len(self.ops)):
            for t in by_first.get(i, []):
                need = _rup(t.nbytes, ALIGN)
                pick = None
                if reuse:
                    for k, (o, sz) in enumerate(free):
                        if sz >= need and (pick is None or sz < free[pick][1]):
                            pick = k
                if pick is not None:
                    o, sz = free.pop(pick)
                    t.off = o
                    if sz > need:
                        free.append((o + need, sz - need))
                else:
                    nxt = (top // WINDOW + 1) * WINDOW                 # start of the next window = its zero page
                    if top + need > nxt:                               # would run into it: leave the rest of this window free
                        if reuse and nxt > top:
                            free.append((top, nxt - top))
                        top = nxt + ZERO_PAGE
                    t.off = top
                    top += need
            for t in expiring.get(i, []):
                if reuse and t.off >= 0:
                    free.append((t.off, _rup(t.nbytes, ALIGN)))
                    free.sort()
                    merged = []
                    for o, sz in free:                                 # (blocks on either side of a zero page are never adjacent)
                        if merged and merged[-1][0] + merged[-1][1] == o:
                            merged[-1] = (merged[-1][0], merged[-1][1] + sz)
                        else:
                            merged.append((o, sz))
                    free = merged
        self.arena_bytes = max(top, ALIGN)
        for t in every:
            if t.off >= 0:
                assert t.off % WINDOW >= ZERO_PAGE and t.off // WINDOW == (t.off + t.nbytes - 1) // WINDOW, t.name
        return self.arena_bytes

    def emit(self):
        arr = (_L.SmapOp * len(self.ops))()
        producer = {}
        for i, op in enumerate(self.ops):
            for t in ([op.out] if op.out is not None else []) + list(op.outs):
                producer[id(t)] = i
        for i, op in enumerate(self.ops):
            o = arr[i]
            C.memset(C.byref(o), 0, C.sizeof(o))
            # lanes: wait for the latest producer on every OTHER lane (ops of one lane are ordered by their stream)
            o.lane = op.lane
            waits = {}
            for t in [op.inp, op.res, op.add1, op.add2, op.aux2] + list(op.aux):
                pi = producer.get(id(t)) if t is not None else None
                if pi is not None and self.ops[pi].lane != op.lane:
                    ln = self.ops[pi].lane
                    waits[ln] = max(waits.get(ln, -1), pi)
            for k in range(4):
                o.wait_op[k] = -1
            for k, pi in enumerate(sorted(waits.values())):
                o.wait_op[k] = pi
            o.n_wait = len(waits)
            o.kind = op.kind
            o.B = self.B
            o.res_off = o.add1_off = o.add2_off = -1
            for k in range(3):
                o.aux_off[k] = -1
            o.in_off = o.out_off = o.w_off = o.bias_off = o.ext_off = -1
            o.precision, o.acc_scale = int(self.x3), 1.0
            p = op.p
            if op.kind == OP_CONV:
                x, y = op.inp, op.out
                o.B = p["frames"]
                o.H, o.W, o.Cin, o.in_stride_c, o.in_c_off = x.H, x.W, p["Cin"], x.C * x.planes, p["in_c_off"]
                o.Ho, o.Wo, o.Cout = y.H, y.W, p["Cout"]
                o.ksize, o.stride, o.pad, o.relu = p["ksize"], p["stride"], p["pad"], p["relu"]
                o.cout_pad, o.out_stride_c, o.out_c_off = p["cout_pad"], y.C * y.planes, 0
                o.acc_scale = p["acc_scale"]
                o.w_pairs = p["w_pairs"]
                o.out_fp32, o.tile = p["out_fp32"], p["tile"]
                if "tail" in p:
                    tl = p["tail"]
                    o.tail_cout, o.tail_cout_pad, o.tail_acc_scale = tl["cout"], tl["cout_pad"], tl["acc_scale"]
                    o.tail_w_off, o.tail_bias_off = tl["w_off"], tl["bias_off"]
                if "head" in p:
                    hd = p["head"]
                    o.head_cin, o.head_acc_scale, o.head_w_off, o.head_bias_off = hd["cin"], hd["acc_scale"], hd["w_off"], hd["bias_off"]
                if "short" in p:
                    o.short_w_off, o.short_acc_scale = p["short"]["w_off"], p["short"]["acc_scale"]
                o.in_off, o.out_off, o.w_off, o.bias_off = x.off, y.off, p["w_off"], p["bias_off"]
                for nm in ("res", "add1", "add2"):
                    t = getattr(op, nm)
                    if t is not None:
                        assert (t.H, t.W, t.C) == (y.H, y.W, y.C) and t.esize == 2, (y.name, nm)
                        setattr(o, nm + "_off", t.off)
                if op.aux:
                    t = op.aux[0]
                    assert t.C == y.C and t.esize == 2
                    o.aux_off[0], o.aux_h[0], o.aux_w[0] = t.off, t.H, t.W
                if "tap" in p:
                    o.tap_n, o.tap_w_off, o.tap_scale = 9, p["tap"]["w_off"], p["tap"]["scale"]
                    o.out_stride_c = 16
                if "cat" in p:
                    t = op.aux2
                    assert t.off // WINDOW == x.off // WINDOW, (y.name, "both inputs of a conv_cat launch must lie in one 4 GiB window")
                    o.in2_off, o.in2_H, o.in2_W, o.in2_C, o.in2_stride_c, o.in2_stride = t.off, t.H, t.W, p["cat"]["cin"], t.C * t.planes, p["cat"]["stride"]
                    if p["cat"].get("relusum"):
                        o.in2_mode, o.in2_acc_scale, o.in2_bias_off = 1, p["cat"]["acc_scale"], p["cat"]["bias_off"]
                if p.get("ksplit", 1) > 1:
                    o.ksplit, o.kpart_off, o.kcount_off = p["ksplit"], op.scratch[0].off, self.kcount.off + 4 * p["kcount_first"]
                for j, (sg, t) in enumerate(zip(p.get("segs", []), op.outs)):
                    assert (t.H, t.W, t.C) == (y.H, y.W, sg["cout"]) and t.esize == 2
                    o.seg_n[j], o.seg_cout[j], o.seg_relu[j], o.seg_acc_scale[j] = sg["n0"], sg["cout"], sg["relu"], sg["acc_scale"]
                    o.seg_out_stride_c[j], o.seg_out_off[j] = t.C * t.planes, t.off
            elif op.kind in (OP_STEM, OP_STEMPOOL):
                y = op.out
                o.H, o.W, o.Cin, o.Ho, o.Wo, o.Cout = self.H, self.W, 3, y.H, y.W, 64
                o.ksize, o.stride, o.pad, o.relu = 7, 2, 3, 1
                o.out_off, o.w_off, o.bias_off = y.off, p["w_off"], p["bias_off"]
                o.acc_scale = p["acc_scale"]
                o.flip_from = self.frames if self.flip_pair is not None else 0
            elif op.kind == OP_MAXPOOL:
                x, y = op.inp, op.out
                o.H, o.W, o.Cin, o.Ho, o.Wo, o.Cout = x.H, x.W, x.C, y.H, y.W, y.C
                o.ksize, o.stride, o.pad = 3, 2, 1
                o.in_off, o.out_off = x.off, y.off
            elif op.kind == OP_UPADD:
                x, y, t = op.inp, op.out, op.aux[0]
                o.H, o.W, o.Cin, o.Ho, o.Wo, o.Cout = x.H, x.W, x.C, y.H, y.W, y.C
                o.relu = p["relu"]
                o.in_off, o.out_off = x.off, y.off
                o.aux_off[0], o.aux_h[0], o.aux_w[0] = t.off, t.H, t.W
            elif op.kind == OP_TAPSUM:
                t = op.aux[0]
                o.H, o.W, o.Cin, o.Ho, o.Wo, o.Cout = t.H, t.W, t.C, self.out_h, self.out_w, 1
                o.aux_off[0], o.aux_h[0], o.aux_w[0] = t.off, t.H, t.W
                o.ext_off, o.bias_off = p["ext_off"], p["bias_off"]
                o.B = self.frames
                o.status_off = self.status_off
            elif op.kind == OP_HEADSUM:
                s0 = op.aux[0]
                assert all(t.C == s0.C and t.esize == 4 for t in op.aux)
                o.H, o.W, o.Cin, o.Ho, o.Wo, o.Cout = s0.H, s0.W, s0.C, self.out_h, self.out_w, p["Cout"]
                o.n_aux = len(op.aux)
                for k, t in enumerate(op.aux):
                    o.aux_off[k], o.aux_h[k], o.aux_w[k] = t.off, t.H, t.W
                o.ext_off = p["ext_off"]
                o.B = self.frames                          # output frames (flip-TTA: the mirrored half is merged in)
                o.status_off = self.status_off
                if p.get("flip_from"):
                    o.flip_from, o.in_c_off, o.w_off = p["flip_from"], p["n_kpt"], p["w_off"]
                if p.get("scale_hms"):
                    o.scale_hms, o.in_c_off = 1, p["n_kpt"]
        return arr

    def blob(self):
        """The allocated schedule as one relocatable byte image (include/smap_hip.h "plan blob": smap_blob_header | smap_op[n]
        | weight section), little-endian: smap_plan_create_from_blob validates it and hands back the buffer sizes and the
        output layout, so that a host with nothing but the header can run SMAP.forward (tests/c/blob_runner.c)."""
        ops = self.emit()
        n_ops = len(self.ops)
        hdr = _L.BlobHeader()
        ops_off = _rup(C.sizeof(hdr), ALIGN)
        ops_bytes = C.sizeof(_L.SmapOp) * n_ops
        w_off = _rup(ops_off + ops_bytes, ALIGN)
        wblob = self.weight_blob().numpy().tobytes()
        hdr.magic, hdr.version, hdr.sizeof_op, hdr.header_bytes = b"SMAPPLN1", _L.BLOB_VERSION, C.sizeof(_L.SmapOp), C.sizeof(hdr)
        hdr.n_ops, hdr.ops_offset = n_ops, ops_off
        hdr.weights_offset, hdr.weights_bytes = w_off, len(wblob)
        hdr.arena_bytes, hdr.out_bytes = self.arena_bytes, self.out_bytes + 4 * self.status_words
        i = hdr.info
        i.frames, i.H, i.W, i.out_h, i.out_w = self.frames, self.H, self.W, self.out_h, self.out_w
        i.n_hms, i.n_det, i.n_root, i.precision = self.kpt_paf, self.paf, 1, int(self.x3)
        i.arena_bytes, i.out_bytes, i.weights_offset, i.weights_bytes = hdr.arena_bytes, hdr.out_bytes, w_off, len(wblob)
        i.hms_off, i.det_off, i.root_off = self.out_layout["hms"][0], self.out_layout["det_d"][0], self.out_layout["root_d"][0]
        i.status_off = self.status_off
        out = bytearray(w_off + len(wblob))
        out[:C.sizeof(hdr)] = bytes(hdr)
        out[ops_off:ops_off + ops_bytes] = bytes(ops)
        out[w_off:] = wblob
        return bytes(out)

    def weight_blob(self):
        blob = torch.zeros((max(self.woff, ALIGN),), dtype=torch.uint8)
        for off, raw in self.wchunks:
            blob[off:off + raw.numel()] = raw
        return blob


# ----------------------------------------------------------------------------- plan cache
# Building a schedule is 2.5-10 s of Python (BN folding in f64, hi/lo splits, weight packing for ~150 launches); its result -- Graph.blob():
# ops + packed weights -- depends only on the checkpoint, the shape, the arithmetic, the tile tables and this file.  With SMAP_PLAN_CACHE=<dir>
# (exps/stage3_root2/test.py defaults it to ~/.cache/smap_amd) BackboneEngine stores the blob there and the next process loads it through
# smap_plan_create_from_blob: < 1 s from the page cache instead of the build.  At most PLAN_CACHE_KEEP blobs are kept (oldest out).
PLAN_CACHE_KEEP = 6


def plan_cache_dir():
    d = os.environ.get("SMAP_PLAN_CACHE", "")
    return None if d in ("", "0") else os.path.expanduser(d)


def state_hash(sd):
    """Content hash of a state dict (keys, shapes, dtypes, bytes): xxh3 when the module is there (~10 GB/s), else blake2b."""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:                                     # pragma: no cover
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    for k in sorted(sd):
        t = sd[k].detach().cpu().contiguous()
        h.update(f"{k}|{tuple(t.shape)}|{t.dtype}|".encode())
        h.update(t.reshape(-1).view(torch.uint8).numpy().data if t.numel() else b"")
    return h.hexdigest()


def _schedule_env():
    """The SMAP_* switches that change what Graph builds (tile overrides, merge policy, split K ...): part of the cache key."""
    skip = ("SMAP_PLAN_CACHE", "SMAP_HIP_LIB", "SMAP_BENCH", "SMAP_CLI", "SMAP_DECODE", "SMAP_STRICT", "SMAP_CHECK", "SMAP_FORCE", "SMAP_GIT",
            "SMAP_MAX_ARENA", "SMAP_MAX_FRAMES", "SMAP_BB_STREAM")
    return sorted((k, v) for k, v in os.environ.items() if k.startswith("SMAP_") and not k.startswith(skip))


def plan_cache_key(sd, B, H, W, stage_num, chl, kpt_paf, paf, precision, flip_pair, scaled_hms):
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.blake2b(digest_size=16)
    h.update(repr((state_hash(sd), B, H, W, stage_num, chl, kpt_paf, paf, precision, tuple(flip_pair) if flip_pair is not None else None,
                   bool(scaled_hms), _L.BLOB_VERSION, C.sizeof(_L.SmapOp), _L.version(), _schedule_env())).encode())
    for f in ("engine.py", "tile_table.json", "tile_table_x3.json"):      # the schedule builder and its tables
        h.update(open(os.path.join(here, f), "rb").read())
    tt = os.environ.get("SMAP_TILE_TABLE_X3") or os.environ.get("SMAP_TILE_TABLE")
    if tt and os.path.exists(tt):
        h.update(open(tt, "rb").read())
    return h.hexdigest()


# ----------------------------------------------------------------------------- engine
class BackboneEngine:
    """Device-resident schedule for one (B, H, W): weights, arena, output buffer, plan."""

    def __init__(self, state_dict, B, H, W, device, stage_num=3, chl=256, kpt_paf=43, paf=14, reuse=True,
                 precision="f16", flip_pair=None, scaled_hms=False):
        self.lib = _L.load()          # fails loudly when libsmap_hip.so is missing
        self.precision = precision
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BackboneEngine needs a ROCm GPU device (no CPU path in smap_amd)")
        sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        self._graph_args = (sd, B, H, W, stage_num, chl, kpt_paf, paf, precision, flip_pair, scaled_hms, reuse)
        self._graph = None
        self.B, self.H, self.W = B, H, W
        self.kpt_paf, self.paf = kpt_paf, paf
        self.from_cache = False
        cdir = plan_cache_dir() if reuse else None         # (reuse=False: debugging layouts, never cached)
        meta = None
        if cdir is not None:
            self.cache_key = plan_cache_key(sd, B, H, W, stage_num, chl, kpt_paf, paf, precision, flip_pair, scaled_hms)
            meta = self._load_cached(cdir)
        if meta is None:
            g = self.graph                                 # builds and allocates the schedule
            meta = dict(arena_bytes=g.arena_bytes, out_h=g.out_h, out_w=g.out_w, out_bytes=g.out_bytes, status_words=g.status_words,
                        lanes=int(g.lanes), flops=g.flops, alg_bytes=g.alg_bytes, n_ops=len(g.ops))
        budget = arena_budget(self.device)
        if meta["arena_bytes"] > budget:
            raise ArenaTooLarge(f"the activation arena of a {B}-frame schedule ({precision}{', flip-TTA' if flip_pair is not None else ''}) takes "
                                f"{meta['arena_bytes'] / 2 ** 30:.2f} GiB, the budget is {budget / 2 ** 30:.2f} GiB (SMAP_MAX_ARENA_BYTES / 90 % of the "
                                "device memory shared by 4 arenas): use smaller launches")
        self.h, self.w = meta["out_h"], meta["out_w"]
        self.n_ops = meta["n_ops"]
        if not self.from_cache:
            g = self.graph
            self.ops = g.emit()
            self.weights = g.weight_blob().to(self.device)
            handle = C.c_void_p()
            with torch.cuda.device(self.device):
                _L.check(self.lib.smap_plan_create(self.ops, self.n_ops, C.byref(handle)), "smap_plan_create")
            self.handle = handle
            if cdir is not None:
                self._store_cached(cdir, meta)
        self.arena = torch.zeros((meta["arena_bytes"],), dtype=torch.uint8, device=self.device)
        if meta["lanes"]:
            _L.check(self.lib.smap_plan_set_lanes(self.handle, 1), "smap_plan_set_lanes")
        self.out_floats = meta["out_bytes"] // 4
        self.status_words = meta["status_words"]
        self.out = self.new_output()                      # default output buffer
        self.hms, self.det_d, self.root_d = self.views(self.out)
        self.flops_per_batch = meta["flops"]
        self.alg_bytes_per_batch = meta["alg_bytes"]      # conv launches only (stem / pool / head sums are < 3 % more)

    @property
    def graph(self):
        """The schedule as Python objects (ops, tensors, weight chunks).  An engine loaded from the plan cache builds it only when somebody
        asks (profiling tools, tests, blob()): the launches themselves need the plan handle alone."""
        if self._graph is None:
            sd, B, H, W, stage_num, chl, kpt_paf, paf, precision, flip_pair, scaled_hms, reuse = self._graph_args
            g = Graph(sd, B, H, W, stage_num, chl, kpt_paf, paf, precision=precision, flip_pair=flip_pair, scaled_hms=scaled_hms)
            g.allocate(reuse=reuse)
            self._graph = g
        return self._graph

    def _load_cached(self, cdir):
        """Plan + weights from <cdir>/<key>.smapplan through smap_plan_create_from_blob, or None (absent / refused: rebuilt and rewritten)."""
        import json
        path = os.path.join(cdir, self.cache_key + ".smapplan")
        if not (os.path.exists(path) and os.path.exists(path + ".json")):
            return None
        try:
            meta = json.load(open(path + ".json"))
            raw = torch.from_file(path, shared=False, size=os.path.getsize(path), dtype=torch.uint8)
            handle, info = C.c_void_p(), _L.BlobInfo()
            with torch.cuda.device(self.device):
                rc = self.lib.smap_plan_create_from_blob(C.c_void_p(raw.data_ptr()), raw.numel(), C.byref(handle), C.byref(info))
            if rc != 0 or info.arena_bytes != meta["arena_bytes"] or info.frames != self.B:
                return None
            self.weights = raw[info.weights_offset:info.weights_offset + info.weights_bytes].to(self.device)
            self.handle, self.ops, self.from_cache = handle, None, True
            os.utime(path)                                  # most recently used
            return meta
        except Exception:                                   # a damaged cache entry is not an error: build the schedule
            return None

    def _store_cached(self, cdir, meta):
        import json
        try:
            os.makedirs(cdir, exist_ok=True)
            path = os.path.join(cdir, self.cache_key + ".smapplan")
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(self.graph.blob())
            os.replace(tmp, path)
            json.dump(meta, open(path + ".json", "w"))
            old = sorted((p for p in (os.path.join(cdir, n) for n in os.listdir(cdir)) if p.endswith(".smapplan")), key=os.path.getmtime)
            for p_ in old[:-PLAN_CACHE_KEEP]:
                for q in (p_, p_ + ".json"):
                    if os.path.exists(q):
                        os.remove(q)
        except OSError:                                     # read-only home, full disk: the cache is an optimisation
            pass

    def sibling(self):
        """A second executor of the same schedule with its own arena (weights and plan shared), so that
        two batches can be in flight on two streams."""
        import copy
        e = copy.copy(self)                # (shares _graph / _graph_args too)
        e.arena = torch.zeros_like(self.arena)
        e.out = e.new_output()
        e.hms, e.det_d, e.root_d = e.views(e.out)
        e._is_sibling = True
        e._parent = self               # the shared plan handle lives as long as any executor of it
        return e

    def new_output(self):
        """A fresh fp32 output buffer (hms | det_d | root_d | status words); pass it to run(out=...) to double-buffer.
        Zero-filled: a partial run(first, count) (debug / trace tools) must not hand back uninitialised memory."""
        return torch.zeros((self.out_floats + self.status_words,), dtype=torch.float32, device=self.device)

    def status(self, out=None):
        """The status word of the last run into `out` (synchronises): bit 0 = a non-finite value reached the output
        maps, i.e. an activation left the fp16 range on the way (split precision has fp16's range, not fp32's)."""
        buf = self.out if out is None else out
        return int(buf[self.out_floats:self.out_floats + 1].view(torch.int32).item())

    def bad_frames(self, out=None):
        """Output frames of the last run into `out` whose maps hold a non-finite value (synchronises): frame f is bit 1 + f % 31 of
        status word f // 31."""
        buf = self.out if out is None else out
        words = buf[self.out_floats:self.out_floats + self.status_words].view(torch.int32).tolist()
        return [f for f in range(self.B) if (words[f // 31] >> (1 + f % 31)) & 1]

    def raise_if_nonfinite(self, out=None):
        if self.status(out) & 1:
            raise RuntimeError("SMAP backbone produced non-finite outputs: an activation exceeded the fp16 range (65504) of the "
                               f"'{self.precision}' arithmetic; see INTEGRATION.md section 5")

    def views(self, out):
        B, n = self.B, self.B * self.h * self.w
        k, p = self.kpt_paf, self.paf
        return (out[:n * k].view(B, k, self.h, self.w), out[n * k:n * (k + p)].view(B, p, self.h, self.w),
                out[n * (k + p):n * (k + p + 1)].view(B, 1, self.h, self.w))

    def __del__(self):
        try:
            if getattr(self, "_is_sibling", False):
                return
            if getattr(self, "handle", None):
                self.lib.smap_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def run(self, imgs, first=0, count=None, out=None):
        """imgs: [B,3,H,W] fp32 contiguous on the device -- or a list / tuple of n (1..8, a divisor of B) such tensors of B / n
        frames each, read where they are (smap_plan_run_inputs: no gather copy in front of the stem).  Returns (hms, det_d,
        root_d) views of the output buffer `out` (default: the engine's own, overwritten by the next run)."""
        parts = list(imgs) if isinstance(imgs, (list, tuple)) else [imgs]
        n = len(parts)
        if not 1 <= n <= _L.MAX_INPUTS or self.B % n:
            raise ValueError(f"{n} input buffers for a schedule of {self.B} frames (1..{_L.MAX_INPUTS} buffers, a divisor of the batch)")
        for t in parts:
            if tuple(t.shape) != (self.B // n, 3, self.H, self.W) or t.dtype != torch.float32 or not t.is_cuda:
                raise ValueError(f"imgs must be float32 GPU tensor(s) [{self.B // n},3,{self.H},{self.W}], got "
                                 f"{tuple(t.shape)} {t.dtype} {t.device}")
        parts = [t.contiguous() for t in parts]
        dev0 = parts[0].device
        if out is not None and (out.dtype != torch.float32 or out.device != dev0 or not out.is_contiguous()
                                or out.numel() < self.out_floats + self.status_words):
            # the maps are followed by the status words: a buffer of the pre-round-3 size would be written past its end
            raise ValueError(f"out must be a contiguous float32 tensor of >= {self.out_floats + self.status_words} elements on {dev0} "
                             f"(BackboneEngine.new_output()), got {out.dtype} x {out.numel()} on {out.device}")
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        outp = C.c_void_p((self.out if out is None else out).data_ptr())
        with torch.cuda.device(self.device):
            if n == 1 or first != 0 or count is not None:
                if n != 1:
                    raise ValueError("partial runs (first / count) take one input buffer")
                count = self.n_ops - first if count is None else count
                _L.check(self.lib.smap_plan_run_range(self.handle, first, count, C.c_void_p(parts[0].data_ptr()),
                                                      C.c_void_p(self.arena.data_ptr()), C.c_void_p(self.weights.data_ptr()), outp, st),
                         "smap_plan_run")
            else:
                ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in parts])
                _L.check(self.lib.smap_plan_run_inputs(self.handle, ptrs, n, C.c_void_p(self.arena.data_ptr()),
                                                       C.c_void_p(self.weights.data_ptr()), outp, st), "smap_plan_run_inputs")
        return (self.hms, self.det_d, self.root_d) if out is None else self.views(out)

    def blob(self):
        """The whole schedule as one relocatable byte image (Graph.blob): what a host without this Python builder loads with
        smap_plan_create_from_blob.  Returns bytes."""
        return self.graph.blob()

    def capture(self, out=None):
        """Record the whole schedule (the ~208 launches of smap_plan_run) into a HIP graph.  Returns
        replay(imgs) -> (hms, det_d, root_d): copies `imgs` into the graph's static input buffer and launches the graph
        on the current stream -- one graph launch instead of ~208 kernel launches, which is what bounds small batches
        (B = 1: 2.9 ms per forward launch by launch).  Arena, weights and `out` are baked into the graph."""
        dev = self.device
        out_t = self.out if out is None else out
        static_in = torch.empty((self.B, 3, self.H, self.W), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.run(static_in, out=out_t)                 # warm-up outside the capture (module load, lazy init)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            self.run(static_in, out=out_t)
        views = (self.hms, self.det_d, self.root_d) if out is None else self.views(out_t)

        def replay(imgs):
            static_in.copy_(imgs, non_blocking=True)
            graph.replay()
            return views
        replay.graph, replay.static_in = graph, static_in      # keep them alive with the closure
        return replay

    def read_tensor(self, name):
        """Debug/test helper: NHWC activation `name` from the arena (valid with reuse=False)."""
        t = next(t for t in self.graph.tensors if t.name == name)
        dt = torch.float16 if t.esize == 2 else torch.float32
        raw = self.arena[t.off:t.off + t.nbytes].view(dt)
        if t.planes == 2:              # split precision: hi + lo as fp32
            raw = raw.view(t.B, t.H, t.W, 2, t.C).float()
            return raw[..., 0, :] + raw[..., 1, :]
        return raw.view(t.B, t.H, t.W, t.C)
