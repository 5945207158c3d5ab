"""GPU parity of the backbone: single conv launches of every tile shape against a plain
PyTorch fp32 conv of the same fp16-rounded operands; the whole schedule at a small shape
against (a) the torch interpretation of the same schedule with the engine's rounding points
and (b) the golden outputs of the imported reference model; size-independent properties at
BASELINE's full 3x512x832."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import make_cfg
from recipe import recipe_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _split(t):
    """fp32 tensor [..., C] -> fp16 [..., 2C] = [hi(C) | lo(C)] (the split-precision storage of csrc/conv.hip X3)."""
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    return torch.cat([hi, lo], -1)


def _run_single_conv(B, H, W, Cin, Cout, k, stride, tile, relu, use_res, use_adds, out_fp32=False,
                     in_stride=None, in_off=0, seed=0, x3=False, x_scale=1.0):
    """One CONV op through smap_plan_run.  x3: split-precision storage (hi/lo planes) and arithmetic; the operands are
    then full fp32 values and the reference is the f64 conv of THOSE."""
    from smap_amd import lib as L
    from smap_amd.engine import TILES, ZERO_PAGE, split_f16
    lib = L.load()
    g = torch.Generator().manual_seed(seed)
    in_stride = in_stride or Cin
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    bn = TILES.get(tile, (128, 64))[1]                 # unknown ids: the plan must reject them
    cout_pad = (Cout + bn - 1) // bn * bn
    c8 = (Cout + 7) // 8 * 8
    K = k * k * Cin
    q = (lambda t: t) if x3 else (lambda t: t.half())          # storage rounding of the operands
    x = q(torch.randn(B, H, W, in_stride, generator=g) * x_scale)
    w = q(torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / K) ** 0.5)
    bias = torch.randn(Cout, generator=g) * x_scale
    res = q(torch.randn(B, Ho, Wo, c8, generator=g) * x_scale) if use_res else None
    a1 = q(torch.randn(B, Ho, Wo, c8, generator=g) * x_scale) if use_adds else None
    a2 = q(torch.randn(B, Ho, Wo, c8, generator=g) * x_scale) if use_adds else None
    acc_scale = 1.0
    if x3:
        hi, lo, acc_scale = split_f16(w.permute(0, 2, 3, 1).reshape(Cout, K).double())
        wk = torch.zeros(2, cout_pad, K, dtype=torch.float16)
        wk[0, :Cout], wk[1, :Cout] = hi, lo
    else:
        wk = torch.zeros(1, cout_pad, K, dtype=torch.float16)
        wk[0, :Cout] = w.permute(0, 2, 3, 1).reshape(Cout, K)
    from smap_amd.engine import pack_conv_weights, tile_family
    w_pairs = seed % 2                                     # both layouts of the 32-half K tiles get exercised
    if tile in TILES and not (tile_family(tile) == "halo" and k != 3):     # (ops the plan must reject keep any bytes)
        wk = pack_conv_weights(wk, tile, x3, k, Cin, pairs=w_pairs)   # weight tiles as contiguous, pre-swizzled blocks (the conv ABI)
    bk = torch.zeros(cout_pad)
    bk[:Cout] = bias
    # weight blob: [wk | bias]; arena: [x | res | a1 | a2 | out]
    al = lambda n: (n + 255) // 256 * 256
    w_bytes = al(wk.numel() * 2)
    blob = torch.zeros(w_bytes + al(bk.numel() * 4), dtype=torch.uint8)
    blob[:wk.numel() * 2] = wk.view(torch.uint8).reshape(-1)
    blob[w_bytes:w_bytes + bk.numel() * 4] = bk.view(torch.uint8).reshape(-1)
    store = _split if x3 else (lambda t: t)
    parts, offs, cur = [x, res, a1, a2], [], ZERO_PAGE    # arena[0:ZERO_PAGE] = zero page
    stored = [store(t) if t is not None else None for t in parts]
    for t in stored:
        offs.append(cur if t is not None else -1)
        cur += al(t.numel() * 2) if t is not None else 0
    out_off = cur
    esz = 4 if out_fp32 else 2
    npl = 2 if (x3 and not out_fp32) else 1
    arena = torch.zeros(out_off + al(B * Ho * Wo * c8 * esz * npl) + 256, dtype=torch.uint8)
    for t, o in zip(stored, offs):
        if t is not None:
            arena[o:o + t.numel() * 2] = t.contiguous().view(torch.uint8).reshape(-1)
    op = L.SmapOp()
    pl_in = 2 if x3 else 1
    op.kind, op.B, op.H, op.W, op.Cin, op.in_stride_c, op.in_c_off = 0, B, H, W, Cin, in_stride * pl_in, in_off
    op.Ho, op.Wo, op.Cout, op.ksize, op.stride, op.pad, op.relu = Ho, Wo, Cout, k, stride, pad, int(relu)
    op.cout_pad, op.out_stride_c, op.out_c_off, op.out_fp32, op.tile = cout_pad, c8 * npl, 0, int(out_fp32), tile
    op.in_off, op.out_off, op.w_off, op.bias_off = offs[0], out_off, 0, w_bytes
    op.res_off, op.add1_off, op.add2_off = offs[1], offs[2], offs[3]
    op.precision, op.acc_scale = int(x3), acc_scale
    op.w_pairs = w_pairs
    for i in range(3):
        op.aux_off[i] = -1
    op.ext_off = -1
    h = C.c_void_p()
    L.check(lib.smap_plan_create(C.byref(op), 1, C.byref(h)), "create")
    arena_d, blob_d = arena.to(DEV), blob.to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena_d.data_ptr()), C.c_void_p(blob_d.data_ptr()), None, st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    raw = arena_d[out_off:out_off + B * Ho * Wo * c8 * esz * npl].cpu()
    if npl == 2:
        got = raw.view(torch.float16).view(B, Ho, Wo, 2, c8).float()
        got = got[..., 0, :] + got[..., 1, :]
    else:
        got = raw.view(torch.float32 if out_fp32 else torch.float16).view(B, Ho, Wo, c8).float()
    # reference: f64 conv of the same stored operands
    xin = x[..., in_off:in_off + Cin].float().permute(0, 3, 1, 2)
    y = F.conv2d(xin.double(), w.double(), bias.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    if use_res:
        y = y + res[..., :Cout].double()
    if relu:
        y = F.relu(y)
    if use_adds:
        y = y + a1[..., :Cout].double() + a2[..., :Cout].double()
    return got, y.float(), Cout


CASES = [
    # B, H,  W,  Cin, Cout, k, s, tile, relu, res, adds
    (2, 16, 24, 64, 256, 1, 1, 0, False, False, False),
    (2, 16, 24, 256, 64, 1, 1, 1, True, False, False),
    (1, 16, 26, 64, 64, 3, 1, 2, True, False, False),      # M = 416: ragged M tile
    (2, 16, 24, 128, 128, 3, 2, 2, True, False, False),
    (2, 16, 24, 256, 256, 1, 1, 0, True, True, True),
    (1, 32, 52, 256, 512, 1, 2, 4, False, False, False),
    (2, 8, 12, 512, 2048, 1, 1, 1, True, True, False),
    (1, 16, 24, 256, 43, 3, 1, 1, False, False, False),
    (1, 16, 24, 256, 14, 3, 1, 3, False, False, False),
    (1, 16, 24, 256, 1, 3, 1, 3, False, False, False),
    (3, 10, 14, 192, 320, 3, 1, 0, True, True, True),      # odd sizes, Cout not a tile multiple
    # deeper LDS-DMA pipelines (tile ids 5..9), incl. K shorter than the pipeline depth
    (2, 16, 24, 64, 256, 1, 1, 5, False, False, False),
    (2, 16, 24, 128, 128, 1, 1, 5, True, True, False),
    (3, 10, 14, 192, 320, 3, 1, 5, True, True, True),
    (2, 16, 24, 256, 64, 3, 1, 6, True, False, False),
    (1, 16, 26, 64, 64, 3, 2, 7, True, False, False),
    (2, 8, 12, 512, 256, 1, 1, 7, True, True, False),
    (1, 16, 24, 256, 14, 3, 1, 8, False, False, False),
    (1, 32, 52, 256, 512, 1, 2, 9, False, False, False),
    # conv.hip with BK = 32 staging, tile ids 20..27
    (2, 16, 24, 64, 256, 1, 1, 20, False, True, False),
    (3, 10, 14, 192, 320, 3, 1, 21, True, True, True),
    (1, 16, 26, 64, 64, 3, 2, 22, True, False, False),
    (2, 8, 12, 512, 256, 1, 1, 23, True, True, True),
    (3, 10, 14, 192, 320, 3, 1, 24, True, True, True),
    (2, 16, 24, 256, 64, 3, 1, 25, True, False, False),
    (1, 16, 24, 256, 43, 3, 1, 26, False, False, False),
    (1, 32, 52, 256, 512, 1, 2, 27, False, True, False),
    # eight-wave workgroups (tile ids 50..54)
    (3, 10, 14, 192, 320, 3, 1, 50, True, True, True),
    (2, 16, 24, 256, 256, 1, 1, 51, True, True, False),
    (2, 16, 24, 128, 128, 3, 2, 52, True, False, False),
    (1, 32, 52, 256, 512, 1, 2, 53, False, True, True),
    (2, 16, 24, 64, 256, 1, 1, 54, True, False, False),
    (1, 16, 26, 64, 320, 3, 1, 54, True, True, True),
    # persistent wave-specialised kernel (convp.hip, tile ids 60..62): register epilogue, loader waves
    (2, 16, 24, 64, 256, 1, 1, 60, False, False, False),
    (3, 10, 14, 192, 320, 3, 1, 60, True, True, True),       # ragged M, Cout not a tile multiple, every epilogue input
    (2, 16, 24, 128, 128, 3, 2, 61, True, False, False),
    (1, 32, 52, 256, 512, 1, 2, 61, False, True, True),
    (8, 64, 104, 64, 256, 1, 1, 62, True, True, False),      # 832 tiles on <= 256 workgroups: the persistent walk
    (2, 8, 12, 512, 256, 1, 1, 62, True, False, True),
    (3, 10, 14, 192, 320, 3, 1, 55, True, True, True),       # 4-stage pipeline
    # halo-tiled 3x3 stride-1 kernel (conv3.hip), tile ids 30..33: ragged pixel tiles in both directions
    (2, 16, 24, 64, 64, 3, 1, 30, True, False, False),
    (3, 10, 14, 192, 320, 3, 1, 31, True, False, False),
    (1, 16, 26, 128, 64, 3, 1, 32, True, False, False),
    (2, 13, 52, 256, 256, 3, 1, 33, False, False, False),
    (1, 16, 24, 256, 43, 3, 1, 30, False, False, False),
    (1, 9, 40, 64, 128, 3, 1, 32, True, False, False),
    (1, 16, 24, 256, 14, 3, 1, 38, False, False, False),
    (2, 10, 40, 128, 1, 3, 1, 39, False, False, False),
    (2, 16, 24, 128, 128, 3, 1, 34, True, False, False),
    (1, 12, 20, 192, 256, 3, 1, 35, True, False, False),
    (1, 16, 36, 64, 64, 3, 1, 36, True, False, False),
    (1, 16, 36, 128, 256, 3, 1, 37, False, False, False),
    # ... with eight waves (tile ids 40..43): 8x16 pixels, 16x16, 16x16 x 64 channels, 8x32
    (2, 16, 24, 128, 128, 3, 1, 40, True, False, False),
    (3, 10, 14, 192, 320, 3, 1, 41, True, False, False),     # ragged in both directions, Cout not a tile multiple
    (1, 16, 24, 256, 43, 3, 1, 42, False, False, False),
    (2, 13, 52, 64, 256, 3, 1, 43, True, False, False),      # one 64-channel chunk
    (1, 32, 52, 256, 256, 3, 1, 41, True, False, False),
    (3, 10, 14, 192, 320, 3, 1, 44, True, False, False),     # staggered schedule
    (2, 13, 52, 64, 256, 3, 1, 45, True, False, False),
    (2, 32, 52, 256, 256, 3, 1, 45, True, False, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[:8])))
def test_single_conv_matches_torch(case):
    got, ref, cout = _run_single_conv(*case, seed=hash(case) % 1000)
    err = (got[..., :cout] - ref).abs()
    tol = 2e-3 * ref.abs().max().item() + 1e-3
    assert torch.isfinite(got).all()
    assert err.max().item() < tol, (err.max().item(), tol, np.unravel_index(err.argmax().item(), err.shape))
    assert not got[..., cout:].any()                       # padded channels are written as zeros


X3_CASES = [
    # split precision (smap_op.precision = 1): the tiles that have an X3 instance, every epilogue flavour
    (2, 16, 24, 64, 256, 1, 1, 20, False, False, False),
    (3, 10, 14, 192, 320, 3, 1, 20, True, True, True),     # odd sizes, Cout not a tile multiple, full epilogue
    (2, 16, 24, 256, 64, 3, 1, 21, True, False, False),
    (1, 16, 26, 64, 64, 3, 2, 22, True, True, False),      # ragged M tile, stride 2
    (2, 8, 12, 512, 256, 1, 1, 23, True, True, True),
    (1, 32, 52, 256, 512, 1, 2, 23, False, False, False),  # strided shortcut
    (2, 8, 12, 2048, 256, 1, 1, 21, True, False, False),   # widest input: zero page must cover the lo plane offset
    (2, 16, 24, 128, 128, 3, 1, 25, True, False, True),
    (1, 16, 24, 256, 43, 3, 1, 27, False, False, False),
    (1, 16, 24, 256, 14, 3, 1, 3, False, False, False),
    (1, 16, 24, 256, 1, 3, 1, 3, False, False, False),
    # BK = 64 and deeper-pipeline instances
    (3, 10, 14, 192, 320, 3, 1, 0, True, True, True),
    (2, 16, 24, 256, 64, 3, 1, 1, True, True, False),
    (1, 16, 26, 64, 64, 3, 2, 2, True, False, False),
    (2, 8, 12, 512, 256, 1, 1, 4, True, False, True),
    (2, 16, 24, 64, 256, 1, 1, 24, False, True, False),      # K shorter than the pipeline depth
    (1, 16, 26, 64, 64, 3, 1, 26, True, False, False),
    (2, 9, 13, 1024, 256, 1, 1, 7, True, True, True),        # tile 7: three 64-half K tiles in flight (batch-1 schedules), long K
    (1, 16, 26, 64, 64, 3, 2, 7, True, False, False),        #   ... and K shorter than the pipeline (one tile per tap)
    (3, 10, 14, 192, 320, 3, 1, 7, True, True, True),
    # eight-wave workgroups
    (3, 10, 14, 192, 320, 3, 1, 50, True, True, True),
    (2, 16, 24, 256, 256, 1, 1, 51, True, True, False),
    (2, 16, 24, 128, 128, 3, 2, 52, True, False, False),
    (1, 32, 52, 256, 512, 1, 2, 53, False, True, True),
    (1, 16, 26, 64, 320, 3, 1, 54, True, True, True),
    # 4-stage pipelines (tile ids 55..57), incl. K shorter than the pipeline
    (3, 10, 14, 192, 320, 3, 1, 55, True, True, True),
    # 256 x 256 with the register epilogue (tile id 56): residual + ReLU only
    (2, 16, 24, 256, 256, 1, 1, 56, True, True, False),      # three M tiles, one N tile
    (3, 10, 14, 192, 320, 3, 1, 56, True, True, False),      # ragged M, Cout not a tile multiple (two N tiles), 3x3 taps
    (1, 32, 52, 256, 512, 1, 2, 56, False, False, False),    # strided shortcut, no ReLU, no residual
    (2, 8, 12, 2048, 1024, 1, 1, 56, True, True, False),     # long K, M < one tile
    (4, 32, 52, 64, 1024, 1, 1, 56, True, False, False),     # K = two tiles (the pipeline's shortest loop), 26 x 4 tiles
    # persistent wave-specialised kernel (convp.hip)
    (2, 16, 24, 64, 256, 1, 1, 60, False, False, False),
    (3, 10, 14, 192, 320, 3, 1, 60, True, True, True),
    (8, 64, 104, 128, 512, 1, 1, 60, True, True, False),     # 832 tiles on <= 256 workgroups, K = 4 tiles of 32
    (2, 16, 24, 128, 128, 3, 2, 61, True, False, False),
    (1, 32, 52, 256, 512, 1, 2, 61, False, True, True),
    (8, 32, 52, 256, 256, 3, 1, 61, True, False, False),     # 3x3 taps through the persistent loader, tiles < workgroups
    (8, 64, 104, 64, 256, 1, 1, 62, True, True, False),
    (2, 8, 12, 2048, 256, 1, 1, 62, True, False, True),
    (2, 16, 24, 256, 64, 1, 1, 63, True, False, False),      # N = 64 tile, 6-stage ring
    # halo-tiled 3x3 (conv3.hip) in split precision: 32-channel chunks, rows = [hi32 | lo32]
    (2, 16, 24, 64, 64, 3, 1, 30, True, False, False),
    (3, 10, 14, 192, 320, 3, 1, 31, True, False, False),
    (1, 16, 26, 128, 64, 3, 1, 32, True, False, False),
    (2, 13, 52, 256, 256, 3, 1, 33, False, False, False),
    (2, 16, 24, 128, 128, 3, 1, 34, True, False, False),
    (1, 12, 20, 192, 256, 3, 1, 35, True, False, False),
    (1, 16, 36, 64, 64, 3, 1, 36, True, False, False),
    (1, 16, 36, 128, 256, 3, 1, 37, False, False, False),
    (1, 16, 24, 256, 14, 3, 1, 38, False, False, False),
    (2, 10, 40, 128, 1, 3, 1, 39, False, False, False),
    # ... with eight waves (tile ids 40..43)
    (2, 16, 24, 128, 128, 3, 1, 40, True, False, False),
    (3, 10, 14, 192, 320, 3, 1, 41, True, False, False),
    (1, 16, 24, 256, 43, 3, 1, 42, False, False, False),
    (2, 13, 52, 64, 256, 3, 1, 43, True, False, False),
    (1, 32, 52, 256, 256, 3, 1, 41, True, False, False),
    (2, 17, 33, 128, 128, 3, 1, 43, True, False, False),
    (3, 10, 14, 192, 320, 3, 1, 44, True, False, False),     # staggered schedule: odd sizes, two n tiles, 6 chunks
    (2, 13, 52, 64, 256, 3, 1, 45, True, False, False),      # two chunks (the shortest pipeline)
    (2, 32, 52, 256, 256, 3, 1, 45, True, False, False),
    (1, 32, 52, 256, 256, 3, 1, 44, False, False, False),
]


@pytest.mark.parametrize("case", X3_CASES, ids=lambda c: "x3-" + "x".join(map(str, c[:8])))
def test_single_conv_split_precision(case):
    """fp16 hi/lo storage + three MFMAs per K step reproduce the fp32 convolution: error at the fp32-roundoff level,
    100x below what one fp16 rounding of the operands would cost (2.4e-4 relative per value)."""
    got, ref, cout = _run_single_conv(*case, seed=hash(case) % 1000, x3=True)
    err = (got[..., :cout] - ref).abs()
    assert torch.isfinite(got).all()
    assert err.max().item() < 3e-6 * ref.abs().max().item() + 1e-6, (err.max().item(), ref.abs().max().item())
    assert not got[..., cout:].any()


def _run_tail(B, H, W, Cin, P, Cout, tile, relu, use_res, use_adds, x3, seed=0):
    """One fused Bottleneck-tail op (csrc/convf.hip): 3x3 Cin -> P (bias, ReLU) then 1x1 P -> Cout (+ res, ReLU, + adds)."""
    from smap_amd import lib as L
    from smap_amd.engine import TAIL_BN, TILES, ZERO_PAGE, pack_halo_rows, split_f16
    lib = L.load()
    g = torch.Generator().manual_seed(seed)
    bn2 = TAIL_BN[tile]
    cout_pad = (Cout + bn2 - 1) // bn2 * bn2
    q = (lambda t: t) if x3 else (lambda t: t.half())
    x = q(torch.randn(B, H, W, Cin, generator=g))
    w3 = q(torch.randn(P, Cin, 3, 3, generator=g) * (1.0 / (9 * Cin)) ** 0.5)
    b3 = torch.randn(P, generator=g) * 0.5
    w1 = q(torch.randn(Cout, P, 1, 1, generator=g) * (1.0 / P) ** 0.5)
    b1 = torch.randn(Cout, generator=g)
    res = q(torch.randn(B, H, W, Cout, generator=g)) if use_res else None
    a1 = q(torch.randn(B, H, W, Cout, generator=g)) if use_adds else None
    a2 = q(torch.randn(B, H, W, Cout, generator=g)) if use_adds else None
    sc3 = sc1 = 1.0
    if x3:
        hi, lo, sc3 = split_f16(w3.permute(0, 2, 3, 1).reshape(P, 9 * Cin).double())
        wk3 = torch.stack([hi, lo])
        hi, lo, sc1 = split_f16(w1.reshape(Cout, P).double())
        wk1 = torch.zeros(2, cout_pad, P, dtype=torch.float16)
        wk1[0, :Cout], wk1[1, :Cout] = hi, lo
    else:
        wk3 = w3.permute(0, 2, 3, 1).reshape(1, P, 9 * Cin)
        wk1 = torch.zeros(1, cout_pad, P, dtype=torch.float16)
        wk1[0, :Cout] = w1.reshape(Cout, P)
    wk3, wk1 = pack_halo_rows(wk3, P, 9, Cin, x3), pack_halo_rows(wk1, bn2, 1, P, x3)
    bk1 = torch.zeros(cout_pad)
    bk1[:Cout] = b1
    al = lambda n: (n + 255) // 256 * 256
    chunks, woffs, cur = [wk3.view(torch.uint8).reshape(-1), b3.view(torch.uint8).reshape(-1), wk1.view(torch.uint8).reshape(-1),
                          bk1.view(torch.uint8).reshape(-1)], [], 0
    for c in chunks:
        woffs.append(cur)
        cur += al(c.numel())
    blob = torch.zeros(cur, dtype=torch.uint8)
    for c, o in zip(chunks, woffs):
        blob[o:o + c.numel()] = c
    store = _split if x3 else (lambda t: t)
    parts, offs, cur = [x, res, a1, a2], [], ZERO_PAGE
    stored = [store(t) if t is not None else None for t in parts]
    for t in stored:
        offs.append(cur if t is not None else -1)
        cur += al(t.numel() * 2) if t is not None else 0
    out_off = cur
    npl = 2 if x3 else 1
    arena = torch.zeros(out_off + al(B * H * W * Cout * 2 * npl) + 256, dtype=torch.uint8)
    for t, o in zip(stored, offs):
        if t is not None:
            arena[o:o + t.numel() * 2] = t.contiguous().view(torch.uint8).reshape(-1)
    op = L.SmapOp()
    op.kind, op.B, op.H, op.W, op.Cin, op.in_stride_c, op.in_c_off = 0, B, H, W, Cin, Cin * npl, 0
    op.Ho, op.Wo, op.Cout, op.ksize, op.stride, op.pad, op.relu = H, W, P, 3, 1, 1, int(relu)
    op.cout_pad, op.out_stride_c, op.out_c_off, op.out_fp32, op.tile = P, Cout * npl, 0, 0, tile
    op.in_off, op.out_off, op.w_off, op.bias_off = offs[0], out_off, woffs[0], woffs[1]
    op.res_off, op.add1_off, op.add2_off = offs[1], offs[2], offs[3]
    op.precision, op.acc_scale = int(x3), sc3
    op.tail_cout, op.tail_cout_pad, op.tail_acc_scale, op.tail_w_off, op.tail_bias_off = Cout, cout_pad, sc1, woffs[2], woffs[3]
    for i in range(3):
        op.aux_off[i] = -1
    op.ext_off = -1
    h = C.c_void_p()
    L.check(lib.smap_plan_create(C.byref(op), 1, C.byref(h)), "create")
    arena_d, blob_d = arena.to(DEV), blob.to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena_d.data_ptr()), C.c_void_p(blob_d.data_ptr()), None, st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    raw = arena_d[out_off:out_off + B * H * W * Cout * 2 * npl].cpu()
    if x3:
        got = raw.view(torch.float16).view(B, H, W, 2, Cout).float()
        got = got[..., 0, :] + got[..., 1, :]
    else:
        got = raw.view(torch.float16).view(B, H, W, Cout).float()
    y = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w3.double(), b3.double(), padding=1))
    if not x3:
        y = y.half().double()                                  # the 3x3's output is an fp16 activation in that mode
    y = F.conv2d(y, w1.double(), b1.double()).permute(0, 2, 3, 1)
    if use_res:
        y = y + res.double()
    if relu:
        y = F.relu(y)
    if use_adds:
        y = y + a1.double() + a2.double()
    return got, y.float()


TAIL_CASES = [
    # B, H,  W,  Cin, P,  Cout, tile, relu, res, adds
    (2, 16, 32, 64, 64, 256, 80, True, True, False),
    (1, 13, 52, 64, 64, 256, 80, True, True, True),          # ragged pixel tiles in both directions
    (3, 10, 14, 128, 64, 192, 80, False, False, False),      # Cout not a chunk multiple, two input chunks
    (2, 16, 32, 64, 64, 256, 81, True, True, True),
    (1, 9, 40, 192, 64, 320, 81, True, True, False),
    (2, 16, 32, 128, 128, 512, 82, True, True, False),
    (1, 13, 52, 128, 128, 448, 82, True, True, True),
]


def _run_block(B, H, W, tile, use_adds, seed=0, mode="full", relu=True, check=True):
    if tile in (92, 93):
        return _run_block_first(B, H, W, tile, seed=seed, mode=mode, check=check)
    """One whole-Bottleneck op (csrc/convb.hip, split precision): relu(W3 relu(W2 * relu(W1 x + b1) + b2) + b3 + x) [+ adds],
    P = 64 planes, C = 256 channels.  `mode` switches parts of the block off so that a failure names the phase
    (tools/debug/convb_probe.py): "residual" (W3 = b3 = 0: out = relu(x)), "no_c1" (W1 = 0: y1 = relu(b1) inside the image),
    "centre_tap" (only the centre tap of the 3x3 is non-zero: no shifted views), "full"."""
    from smap_amd import lib as L
    from smap_amd.engine import TAIL_BN, ZERO_PAGE, pack_halo_rows, pack_rows16, split_f16
    lib = L.load()
    P, Cc = (128, 512) if tile == 94 else (64, 256)           # 94: csrc/convc.hip (layer2's width)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cc, generator=g)
    w1 = torch.randn(P, Cc, 1, 1, generator=g) * (1.0 / Cc) ** 0.5
    b1 = torch.randn(P, generator=g) * 0.5
    w3 = torch.randn(P, P, 3, 3, generator=g) * (1.0 / (9 * P)) ** 0.5
    b3 = torch.randn(P, generator=g) * 0.5
    wt = torch.randn(Cc, P, 1, 1, generator=g) * (1.0 / P) ** 0.5
    bt = torch.randn(Cc, generator=g)
    a1 = torch.randn(B, H, W, Cc, generator=g) if use_adds else None
    a2 = torch.randn(B, H, W, Cc, generator=g) if use_adds else None
    if mode == "residual":
        wt, bt = wt * 0, bt * 0
    elif mode == "no_c1":
        w1 = w1 * 0
    elif mode == "centre_tap":
        m = torch.zeros(3, 3)
        m[1, 1] = 1
        w3 = w3 * m
    bn2 = TAIL_BN[tile]
    hi, lo, sc1 = split_f16(w1.reshape(P, Cc).double())
    wk1 = pack_rows16(torch.stack([hi, lo])) if P == 64 else pack_halo_rows(torch.stack([hi, lo]), P, 1, Cc, True)
    hi, lo, sc3 = split_f16(w3.permute(0, 2, 3, 1).reshape(P, 9 * P).double())
    wk3 = pack_halo_rows(torch.stack([hi, lo]), P, 9, P, True)
    hi, lo, sct = split_f16(wt.reshape(Cc, P).double())
    wkt = pack_halo_rows(torch.stack([hi, lo]), bn2, 1, P, True)
    al = lambda n: (n + 255) // 256 * 256
    raw8 = lambda t: t.contiguous().view(torch.uint8).reshape(-1)
    chunks, woffs, cur = [raw8(wk3), raw8(b3), raw8(wkt), raw8(bt), raw8(wk1), raw8(b1)], [], 0
    for c in chunks:
        woffs.append(cur)
        cur += al(c.numel())
    blob = torch.zeros(cur, dtype=torch.uint8)
    for c, o in zip(chunks, woffs):
        blob[o:o + c.numel()] = c
    parts, offs, cur = [x, a1, a2], [], ZERO_PAGE
    stored = [_split(t) if t is not None else None for t in parts]
    for t in stored:
        offs.append(cur if t is not None else -1)
        cur += al(t.numel() * 2) if t is not None else 0
    out_off = cur
    arena = torch.zeros(out_off + al(B * H * W * Cc * 4) + 256, dtype=torch.uint8)
    for t, o in zip(stored, offs):
        if t is not None:
            arena[o:o + t.numel() * 2] = raw8(t)
    op = L.SmapOp()
    op.kind, op.B, op.H, op.W, op.Cin, op.in_stride_c, op.in_c_off = 0, B, H, W, P, Cc * 2, 0
    op.Ho, op.Wo, op.Cout, op.ksize, op.stride, op.pad, op.relu = H, W, P, 3, 1, 1, int(relu)
    op.cout_pad, op.out_stride_c, op.out_c_off, op.out_fp32, op.tile = P, Cc * 2, 0, 0, tile
    op.in_off, op.out_off, op.w_off, op.bias_off = offs[0], out_off, woffs[0], woffs[1]
    op.res_off, op.add1_off, op.add2_off = offs[0], offs[1], offs[2]
    op.precision, op.acc_scale = 1, sc3
    op.tail_cout, op.tail_cout_pad, op.tail_acc_scale, op.tail_w_off, op.tail_bias_off = Cc, Cc, sct, woffs[2], woffs[3]
    op.head_cin, op.head_acc_scale, op.head_w_off, op.head_bias_off = Cc, sc1, woffs[4], woffs[5]
    for i in range(3):
        op.aux_off[i] = -1
    op.ext_off = -1
    h = C.c_void_p()
    L.check(lib.smap_plan_create(C.byref(op), 1, C.byref(h)), "create")
    arena_d, blob_d = arena.to(DEV), blob.to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena_d.data_ptr()), C.c_void_p(blob_d.data_ptr()), None, st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    if not check:                                            # tools/trace_convb.py: the launch is what matters
        return None, None
    got = arena_d[out_off:out_off + B * H * W * Cc * 4].cpu().view(torch.float16).view(B, H, W, 2, Cc).float()
    got = got[..., 0, :] + got[..., 1, :]
    xin = x.double().permute(0, 3, 1, 2)
    y = F.relu(F.conv2d(xin, w1.double(), b1.double()))
    y = F.relu(F.conv2d(y, w3.double(), b3.double(), padding=1))
    y = (F.conv2d(y, wt.double(), bt.double()) + xin).permute(0, 2, 3, 1)
    if relu:
        y = F.relu(y)
    if use_adds:
        y = y + a1.double() + a2.double()
    return got, y.float()


def _run_block_first(B, H, W, tile, seed=0, mode="full", check=True):
    """The FIRST block of layer1 in one launch (csrc/convb.hip, tile ids 92, 93): 64 input channels,
    relu(W3 relu(W2 * relu(W1 x + b1) + b2) + b3 + Wd x + bd).  Modes: "residual" (W3 = 0: out = relu(Wd x + bd + b3)), "no_c1",
    "centre_tap", "full"."""
    from smap_amd import lib as L
    from smap_amd.engine import ZERO_PAGE, pack_halo_rows, split_f16
    lib = L.load()
    P, Cin, Cc = 64, 64, 256
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g)
    w1 = torch.randn(P, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5
    b1 = torch.randn(P, generator=g) * 0.5
    w3 = torch.randn(P, P, 3, 3, generator=g) * (1.0 / (9 * P)) ** 0.5
    b3 = torch.randn(P, generator=g) * 0.5
    wt = torch.randn(Cc, P, 1, 1, generator=g) * (1.0 / P) ** 0.5
    bt = torch.randn(Cc, generator=g)
    wd = torch.randn(Cc, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5
    bd = torch.randn(Cc, generator=g)
    if mode == "residual":
        wt = wt * 0
    elif mode == "no_c1":
        w1 = w1 * 0
    elif mode == "centre_tap":
        m = torch.zeros(3, 3)
        m[1, 1] = 1
        w3 = w3 * m
    pk = lambda w2d, bn, taps, cin: pack_halo_rows(torch.stack(split_f16(w2d.double())[:2]), bn, taps, cin, True)
    sc1, sc3, sct, scd = (split_f16(t.double())[2] for t in (w1.reshape(P, Cin), w3.permute(0, 2, 3, 1).reshape(P, 9 * P), wt.reshape(Cc, P), wd.reshape(Cc, Cin)))
    wk1, wk3 = pk(w1.reshape(P, Cin), P, 1, Cin), pk(w3.permute(0, 2, 3, 1).reshape(P, 9 * P), P, 9, P)
    wkt, wkd = pk(wt.reshape(Cc, P), 64, 1, P), pk(wd.reshape(Cc, Cin), 64, 1, Cin)
    al = lambda n: (n + 255) // 256 * 256
    raw8 = lambda t: t.contiguous().view(torch.uint8).reshape(-1)
    chunks, woffs, cur = [raw8(wk3), raw8(b3), raw8(wkt), raw8(bt + bd), raw8(wk1), raw8(b1), raw8(wkd)], [], 0
    for c in chunks:
        woffs.append(cur)
        cur += al(c.numel())
    blob = torch.zeros(cur, dtype=torch.uint8)
    for c, o in zip(chunks, woffs):
        blob[o:o + c.numel()] = c
    xs = _split(x)
    in_off = ZERO_PAGE
    out_off = in_off + al(xs.numel() * 2)
    arena = torch.zeros(out_off + al(B * H * W * Cc * 4) + 256, dtype=torch.uint8)
    arena[in_off:in_off + xs.numel() * 2] = raw8(xs)
    op = L.SmapOp()
    op.kind, op.B, op.H, op.W, op.Cin, op.in_stride_c, op.in_c_off = 0, B, H, W, P, Cin * 2, 0
    op.Ho, op.Wo, op.Cout, op.ksize, op.stride, op.pad, op.relu = H, W, P, 3, 1, 1, 1
    op.cout_pad, op.out_stride_c, op.out_c_off, op.out_fp32, op.tile = P, Cc * 2, 0, 0, tile
    op.in_off, op.out_off, op.w_off, op.bias_off = in_off, out_off, woffs[0], woffs[1]
    op.res_off = op.add1_off = op.add2_off = -1
    op.precision, op.acc_scale = 1, sc3
    op.tail_cout, op.tail_cout_pad, op.tail_acc_scale, op.tail_w_off, op.tail_bias_off = Cc, Cc, sct, woffs[2], woffs[3]
    op.head_cin, op.head_acc_scale, op.head_w_off, op.head_bias_off = Cin, sc1, woffs[4], woffs[5]
    op.short_w_off, op.short_acc_scale = woffs[6], scd
    for i in range(3):
        op.aux_off[i] = -1
    op.ext_off = -1
    h = C.c_void_p()
    L.check(lib.smap_plan_create(C.byref(op), 1, C.byref(h)), "create")
    arena_d, blob_d = arena.to(DEV), blob.to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena_d.data_ptr()), C.c_void_p(blob_d.data_ptr()), None, st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    if not check:
        return None, None
    got = arena_d[out_off:out_off + B * H * W * Cc * 4].cpu().view(torch.float16).view(B, H, W, 2, Cc).float()
    got = got[..., 0, :] + got[..., 1, :]
    xin = x.double().permute(0, 3, 1, 2)
    y = F.relu(F.conv2d(xin, w1.double(), b1.double()))
    y = F.relu(F.conv2d(y, w3.double(), b3.double(), padding=1))
    y = F.relu(F.conv2d(y, wt.double(), bt.double()) + F.conv2d(xin, wd.double(), bd.double())).permute(0, 2, 3, 1)
    return got, y.float()


BLOCK_CASES = [
    # B, H,  W,  tile, adds
    (2, 16, 32, 90, False),
    (1, 13, 52, 90, True),           # ragged pixel tiles in both directions (4 x 16 tiles)
    (3, 10, 14, 90, False),          # narrower than one tile
    (2, 16, 32, 91, False),
    (1, 13, 52, 91, True),           # ragged 8 x 16 tiles
    (3, 10, 14, 91, True),
    (1, 32, 208, 91, False),         # a full row of 13 tiles
    # first block of a layer: 64 input channels, 1x1 shortcut conv (tile ids 92, 93)
    (2, 16, 32, 92, False),
    (1, 13, 52, 92, False),
    (3, 10, 14, 93, False),
    (2, 16, 32, 93, False),
    (1, 13, 52, 93, False),
    # 128 planes / 512 channels (csrc/convc.hip, tile id 94: layer2's identity blocks), 8 x 16 tiles, eight waves
    (2, 16, 32, 94, False),
    (1, 13, 52, 94, True),           # ragged tiles, skip adds
    (3, 10, 14, 94, True),           # narrower than one tile
    (1, 24, 104, 94, False),         # a full row of 7 tiles (the last one ragged)
]


@pytest.mark.parametrize("mode", ["full", "residual", "no_c1", "centre_tap"])
@pytest.mark.parametrize("case", BLOCK_CASES, ids=lambda c: "x".join(map(str, c[:4])) + ("+adds" if c[4] else ""))
def test_whole_bottleneck_launch(case, mode):
    """csrc/convb.hip: the three convs of an identity Bottleneck + residual in one launch against the f64 evaluation of the
    same block on the same split-precision operands; the partial modes name the phase when something is off."""
    got, ref = _run_block(*case, seed=(hash(case) + len(mode)) % 1000, mode=mode)
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    tol = 3e-6 * ref.abs().max().item() + 1e-6
    assert err.max().item() < tol, (err.max().item(), tol, np.unravel_index(err.argmax().item(), err.shape))


@pytest.mark.parametrize("spec", ["64:90", "64:91", "64:91+64:93", "64:90+64:92", "64:91,128:94+64:93"])
def test_small_schedule_with_whole_bottleneck_launches(golden_dir, small, monkeypatch, spec):
    """Identity Bottlenecks of layer1 as ONE launch each (csrc/convb.hip): every stored tensor against the f64 interpretation
    of the SAME schedule, and the outputs against the golden outputs of the reference model."""
    from smap_amd.engine import BackboneEngine, Graph
    from oracle.graph_interp import run_graph
    _, sd = small
    z = np.load(f"{golden_dir}/backbone_small.npz")
    x = torch.from_numpy(z["x"])
    spec, _, first = spec.partition("+")
    monkeypatch.setenv("SMAP_BLOCK", spec)
    monkeypatch.setenv("SMAP_BLOCK_FIRST", first)
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False, precision="x3")
    n2 = 9 if "128:" in spec else 0                          # layer2: three identity blocks per stage
    assert sum(1 for op in eng.graph.ops if op.kind == 0 and "head" in op.p) == (9 if first else 6) + n2
    assert sum(1 for op in eng.graph.ops if op.kind == 0 and "short" in op.p) == (3 if first else 0)
    assert not any(t.name.endswith((".c1", ".c2")) for t in eng.graph.tensors if ".layer1.1" in t.name or ".layer1.2" in t.name)
    outs = [o.cpu() for o in eng.run(x.to(DEV))]
    torch.cuda.synchronize()
    g = Graph(sd, 2, 64, 96, keep_ref=True, precision="x3")
    with torch.no_grad():
        *ref, T = run_graph(g, x.double(), quantize=False, keep=True)
    worst = []
    for t in eng.graph.tensors:
        got = eng.read_tensor(t.name).cpu().double().permute(0, 3, 1, 2)
        want = T[t.name].double()
        c = want.shape[1]
        worst.append(((got[:, :c] - want).abs().max().item() / (want.abs().max().item() + 1e-6), t.name))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-5, worst[:5]
    for a, k in zip(outs, ("hms", "det_d", "root_d")):
        assert np.abs(a.numpy() - z[k]).max() < 2e-5 * np.abs(z[k]).max(), k


@pytest.mark.parametrize("x3", [False, True], ids=["f16", "x3"])
@pytest.mark.parametrize("case", TAIL_CASES, ids=lambda c: "x".join(map(str, c[:7])))
def test_fused_bottleneck_tail(case, x3):
    got, ref = _run_tail(*case, x3=x3, seed=hash(case) % 1000)
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    tol = 3e-6 * ref.abs().max().item() + 1e-6 if x3 else 2e-3 * ref.abs().max().item() + 1e-3
    assert err.max().item() < tol, (err.max().item(), tol, np.unravel_index(err.argmax().item(), err.shape))


def test_split_precision_keeps_fp16_subnormal_lo_parts():
    """Activations around 2^-6: their lo parts (~2^-18) are SUBNORMAL fp16 numbers.  The split scheme relies on the matrix
    cores taking subnormal fp16 inputs as they are; flushing them would cost 2^-12 relative error on such values."""
    got, ref, cout = _run_single_conv(2, 16, 24, 256, 256, 1, 1, 20, False, False, False, seed=3, x3=True, x_scale=2.0 ** -6)
    err = (got[..., :cout] - ref).abs().max().item()
    assert err < 1e-5 * ref.abs().max().item(), (err, ref.abs().max().item())


def test_split_precision_fp32_out_and_channel_slice():
    got, ref, cout = _run_single_conv(1, 16, 24, 256, 43, 3, 1, 21, False, False, False, out_fp32=True,
                                      in_stride=768, in_off=256, seed=5, x3=True)
    assert (got[..., :cout] - ref).abs().max().item() < 3e-6 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("tile", [30, 36, 38, 42])
def test_halo_conv_split_precision_fp32_out_and_channel_slice(tile):
    cout = 14 if tile == 38 else 43
    got, ref, cout = _run_single_conv(1, 16, 24, 256, cout, 3, 1, tile, False, False, False, out_fp32=True,
                                      in_stride=768, in_off=512, seed=7, x3=True)
    assert (got[..., :cout] - ref).abs().max().item() < 3e-6 * ref.abs().max().item() + 1e-6


def test_single_conv_fp32_out_and_channel_slice():
    got, ref, cout = _run_single_conv(1, 16, 24, 256, 43, 3, 1, 1, False, False, False, out_fp32=True,
                                      in_stride=768, in_off=256, seed=5)
    assert (got[..., :cout] - ref).abs().max().item() < 2e-4 * ref.abs().max().item() + 1e-4


@pytest.mark.parametrize("tile", [30, 32])
def test_halo_conv_fp32_out_and_channel_slice(tile):
    got, ref, cout = _run_single_conv(1, 16, 24, 256, 43, 3, 1, tile, False, False, False, out_fp32=True,
                                      in_stride=768, in_off=256, seed=6)
    assert (got[..., :cout] - ref).abs().max().item() < 2e-4 * ref.abs().max().item() + 1e-4


def test_halo_conv_rejects_fused_epilogues():
    # the halo kernel has no residual / addend path: the plan must refuse such an op, not mis-run it
    from smap_amd.lib import SmapError
    with pytest.raises(SmapError):
        _run_single_conv(1, 16, 24, 64, 64, 3, 1, 30, True, True, False)
    with pytest.raises(SmapError):
        _run_single_conv(1, 16, 24, 64, 64, 1, 1, 30, True, False, False)
    with pytest.raises(SmapError):                     # tile ids of kernels that are no longer in the product build
        _run_single_conv(1, 16, 24, 64, 256, 1, 1, 45, True, False, False)
    with pytest.raises(SmapError):
        _run_single_conv(1, 16, 24, 64, 256, 1, 1, 12, True, False, False)
    with pytest.raises(SmapError):                     # split precision only on the tiles that have an X3 instance
        _run_single_conv(1, 16, 24, 64, 64, 3, 1, 5, True, False, False, x3=True)
    with pytest.raises(SmapError):                     # halo kernel in split precision: still plain 3x3 only
        _run_single_conv(1, 16, 24, 64, 64, 3, 1, 30, True, True, False, x3=True)


@pytest.fixture(scope="module")
def small():
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    sd = recipe_state_dict(net.state_dict())
    net.load_state_dict(sd)
    return net, sd


def test_small_schedule_every_tensor_vs_interpreter(golden_dir, small):
    from smap_amd.engine import BackboneEngine, Graph
    from oracle.graph_interp import run_graph
    _, sd = small
    z = np.load(f"{golden_dir}/backbone_small.npz")
    x = torch.from_numpy(z["x"])
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False)
    outs = [o.cpu() for o in eng.run(x.to(DEV))]
    torch.cuda.synchronize()
    g = Graph(sd, 2, 64, 96, keep_ref=True)
    with torch.no_grad():
        *ref, T = run_graph(g, x, quantize=True, keep=True)
    worst = []
    for t in eng.graph.tensors:
        got = eng.read_tensor(t.name).cpu().float().permute(0, 3, 1, 2)
        want = T[t.name]
        c = want.shape[1]
        e = (got[:, :c] - want).abs().max().item()
        m = want.abs().max().item()
        worst.append((e / (m + 1e-6), t.name, e, m))
    worst.sort(reverse=True)
    # fp16 storage: one half-ulp (2^-11) per op, errors compound over ~150 ops in depth
    assert worst[0][0] < 2e-2, worst[:5]
    for a, b, k in zip(outs, ref, ("hms", "det_d", "root_d")):
        assert (a - b).abs().max().item() < 5e-3 * b.abs().max().item(), k
        # and against the imported reference model itself
        assert np.abs(a.numpy() - z[k]).max() < 1e-2 * np.abs(z[k]).max(), k


@pytest.mark.parametrize("tile", [None, "22", "20"], ids=["heuristic", "all64x64", "all128x128"])
def test_small_schedule_split_precision_every_tensor(golden_dir, small, monkeypatch, tile):
    """precision "x3" (fp16 hi/lo pairs, three MFMAs per K step): every tensor of the schedule against the fp32 torch
    interpretation of the same schedule, and the outputs against the IMPORTED reference model's golden outputs, at the
    fp32-roundoff level -- SURVEY.md 7 step 4's "<= 1e-4 relative" with two orders of magnitude to spare."""
    from smap_amd.engine import BackboneEngine, Graph
    from oracle.graph_interp import run_graph
    _, sd = small
    if tile:
        monkeypatch.setenv("SMAP_X3_TILE", tile)
    monkeypatch.setenv("SMAP_BLOCK", "")                 # layer by layer: this test is about the tiles of the single-conv kernels (the
    monkeypatch.setenv("SMAP_BLOCK_FIRST", "")           # default schedule's whole-block launches: test_small_schedule_with_whole_bottleneck_launches)
    z = np.load(f"{golden_dir}/backbone_small.npz")
    x = torch.from_numpy(z["x"])
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False, precision="x3")
    from smap_amd.engine import X3_TILES
    assert all(op.p["tile"] in X3_TILES + (3,) + tuple(range(30, 46)) for op in eng.graph.ops if op.kind == 0)
    outs = [o.cpu() for o in eng.run(x.to(DEV))]
    torch.cuda.synchronize()
    g = Graph(sd, 2, 64, 96, keep_ref=True)
    with torch.no_grad():
        *ref, T = run_graph(g, x.double(), quantize=False, keep=True)
    worst = []
    for t in eng.graph.tensors:
        got = eng.read_tensor(t.name).cpu().double().permute(0, 3, 1, 2)
        want = T[t.name].double()
        c = want.shape[1]
        e = (got[:, :c] - want).abs().max().item()
        worst.append((e / (want.abs().max().item() + 1e-6), t.name))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-5, worst[:5]
    for a, k in zip(outs, ("hms", "det_d", "root_d")):
        assert np.abs(a.numpy() - z[k]).max() < 2e-5 * np.abs(z[k]).max(), k


@pytest.mark.parametrize("precision,tol", [("x3", 2e-5), ("f16", 2e-2)])
@pytest.mark.parametrize("merge", ["1", "2", "0"], ids=["merged_1x1", "all_merges", "one_launch_each"])
def test_small_schedule_merged_shared_input_launches(golden_dir, small, monkeypatch, merge, precision, tol):
    """The shared-input 1x1 convs of every Upsample_unit (smap.py:210-241: u_skip | skip1 on x; skip2 | cross_conv | res_conv1 | the
    next unit's up_conv on `out`) as ONE launch with one output tensor per conv (smap_op.seg_*) -- and, SMAP_MERGE_1X1=0, as one
    launch each: every tensor of both schedules against the torch interpretation, the outputs against the imported reference."""
    from smap_amd.engine import BackboneEngine, Graph
    from oracle.graph_interp import run_graph
    _, sd = small
    monkeypatch.setenv("SMAP_MERGE_1X1", merge)
    monkeypatch.setenv("SMAP_SKIPSUM", "0")              # (the skip convs as segments of these launches; the default -- skip1 + skip2 as one launch -- is below)
    z = np.load(f"{golden_dir}/backbone_small.npz")
    x = torch.from_numpy(z["x"])
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False, precision=precision)
    assert sum(len(op.outs) for op in eng.graph.ops) == {"2": 18, "1": 12, "0": 0}[merge]
    outs = [o.cpu() for o in eng.run(x.to(DEV))]
    torch.cuda.synchronize()
    g = Graph(sd, 2, 64, 96, keep_ref=True, precision=precision)
    with torch.no_grad():
        *ref, T = run_graph(g, x.double() if precision == "x3" else x, quantize=precision == "f16", keep=True)
    worst = []
    for t in eng.graph.tensors:
        got = eng.read_tensor(t.name).cpu().double().permute(0, 3, 1, 2)
        want = T[t.name].double()
        e = (got[:, :want.shape[1]] - want).abs().max().item()
        worst.append((e / (want.abs().max().item() + 1e-6), t.name))
    worst.sort(reverse=True)
    assert worst[0][0] < tol, worst[:5]
    for a, k in zip(outs, ("hms", "det_d", "root_d")):
        assert np.abs(a.numpy() - z[k]).max() < (2e-5 if precision == "x3" else 1e-2) * np.abs(z[k]).max(), k


@pytest.mark.parametrize("precision,tol", [("x3", 2e-5), ("f16", 2e-2)])
@pytest.mark.parametrize("splitk,x3tile", [("", None), ("0", None), ("3", None), ("3", "2"), ("16", "20"), ("5", "7")],
                         ids=["rule", "off", "three_parts", "three_parts_tile2", "sixteen_parts_tile20", "five_parts_tile7"])
def test_small_schedule_split_k_launches(golden_dir, small, monkeypatch, splitk, x3tile, precision, tol):
    """Split K (include/smap_hip.h smap_op.ksplit): S workgroups per output tile, each over 1/S of the K tiles, partial tiles summed in
    a fixed order by the last one to arrive -- what the batch-1 schedule runs on its 32x52 / 16x26 levels.  The small schedule has
    few output tiles everywhere, so the engine's rule splits most of its launches (1x1, 3x3, strided, residual / skip-add epilogues,
    fused bilinear add, merged multi-output launches); forced to 3 parts (K tiles do not divide evenly) and to the maximum: every
    tensor against the torch interpretation, outputs against the imported reference, and two runs bit for bit."""
    from smap_amd.engine import BackboneEngine, Graph
    from oracle.graph_interp import run_graph
    _, sd = small
    if splitk:
        monkeypatch.setenv("SMAP_SPLITK", splitk)
    if precision == "f16" and splitk not in ("", "3"):
        pytest.skip("fp16 storage: the rule and one forced case (the suite's run time)")
    if x3tile:
        if precision != "x3":
            pytest.skip("SMAP_X3_TILE forces split-precision tiles")
        monkeypatch.setenv("SMAP_X3_TILE", x3tile)       # the other tiles with a split-K instance (2 / 7: what batch 1 at 512x832 runs)
        monkeypatch.setenv("SMAP_DEEP_TILE", "0")        # (keep tile 2 where it is asked for: the rule would turn these small launches into tile 7)
    monkeypatch.setenv("SMAP_BLOCK", "")                 # layer by layer: more launches of the kernel under test
    monkeypatch.setenv("SMAP_BLOCK_FIRST", "")
    z = np.load(f"{golden_dir}/backbone_small.npz")
    x = torch.from_numpy(z["x"])
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=True, precision=precision)
    n_split = sum(1 for op in eng.graph.ops if op.p.get("ksplit", 1) > 1)
    assert n_split == 0 if splitk == "0" else n_split >= (1 if splitk == "" else 30), n_split     # (the rule splits K >= 2048 only)
    if splitk in ("3", "5", "16"):
        assert max(op.p.get("ksplit", 1) for op in eng.graph.ops) == int(splitk)
    if x3tile:
        assert sum(1 for op in eng.graph.ops if op.p.get("ksplit", 1) > 1 and op.p["tile"] == int(x3tile)) >= 30
    runs = []
    for _ in range(3):
        out = eng.new_output()
        eng.run(x.to(DEV), out=out)
        torch.cuda.synchronize()
        runs.append(out.cpu())
    assert torch.equal(runs[0].view(torch.int32), runs[1].view(torch.int32)) and torch.equal(runs[0].view(torch.int32), runs[2].view(torch.int32))
    eng2 = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False, precision=precision)
    outs = [o.cpu() for o in eng2.run(x.to(DEV))]
    torch.cuda.synchronize()
    g = Graph(sd, 2, 64, 96, keep_ref=True, precision=precision)
    with torch.no_grad():
        *ref, T = run_graph(g, x.double() if precision == "x3" else x, quantize=precision == "f16", keep=True)
    worst = []
    for t in eng2.graph.tensors:
        got = eng2.read_tensor(t.name).cpu().double().permute(0, 3, 1, 2)
        want = T[t.name].double()
        e = (got[:, :want.shape[1]] - want).abs().max().item()
        worst.append((e / (want.abs().max().item() + 1e-6), t.name))
    worst.sort(reverse=True)
    assert worst[0][0] < tol, worst[:5]
    for a, k in zip(outs, ("hms", "det_d", "root_d")):
        assert np.abs(a.numpy() - z[k]).max() < (2e-5 if precision == "x3" else 1e-2) * np.abs(z[k]).max(), k
    assert torch.equal(torch.cat([o.reshape(-1) for o in outs]), runs[0][:sum(o.numel() for o in outs)])     # reuse=True arena: same bits


@pytest.mark.parametrize("precision", ["x3", "f16"])
def test_lanes_fork_the_head_chains_and_change_nothing(small, monkeypatch, precision):
    """smap_op.lane / smap_plan_set_lanes: the 3x3 head of an Upsample_unit runs on a forked stream beside the next unit, the three
    heads of up4 and the three head sums side by side (model/smap.py:219-229 are independent of the unit that follows).  Same kernels,
    same operands: the outputs must equal the single-stream run BIT FOR BIT, launched directly, replayed from a captured graph (the
    lanes become parallel branches) and ten times over (a missing wait shows as a difference sooner or later)."""
    from smap_amd.engine import BackboneEngine
    _, sd = small
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(5)).to(DEV)
    monkeypatch.setenv("SMAP_LANES", "0")
    e0 = BackboneEngine(sd, 2, 64, 96, DEV, precision=precision)
    assert not e0.graph.lanes and all(op.lane == 0 for op in e0.graph.ops)
    o0 = e0.new_output()
    e0.run(x, out=o0)
    torch.cuda.synchronize()
    monkeypatch.setenv("SMAP_LANES", "1")
    e1 = BackboneEngine(sd, 2, 64, 96, DEV, precision=precision)
    ops = e1.graph.emit()
    assert e1.graph.lanes and sorted({op.lane for op in e1.graph.ops}) == [0, 1, 2]
    assert sum(1 for o in ops if o.n_wait) >= 5 and all(ops[o.wait_op[k]].lane != o.lane for o in ops for k in range(o.n_wait))
    for _ in range(10):
        o1 = e1.new_output()
        e1.run(x, out=o1)
        torch.cuda.synchronize()
        assert torch.equal(o0.view(torch.int32), o1.view(torch.int32))
    og = e1.new_output()
    replay = e1.capture(og)
    for _ in range(5):
        og.zero_()
        replay(x)
        torch.cuda.synchronize()
        assert torch.equal(o0.view(torch.int32), og.view(torch.int32))
    s2 = torch.cuda.Stream()                              # a sibling (PosePipeline depth 2) shares the plan and its side streams
    e2 = e1.sibling()
    o2 = e2.new_output()
    with torch.cuda.stream(s2):
        e2.run(x, out=o2)
    o3 = e1.new_output()
    e1.run(x, out=o3)
    torch.cuda.synchronize()
    assert torch.equal(o0.view(torch.int32), o2.view(torch.int32)) and torch.equal(o0.view(torch.int32), o3.view(torch.int32))


def test_split_k_batch_1_full_size_many_runs_bit_for_bit(monkeypatch):
    """The batch-1 schedule at 512x832 (configs[1]: 34 split-K launches, partial tiles crossing XCDs through agent-scope accesses, no
    fence) 300 times on two alternating streams: every run must reproduce the first BIT FOR BIT (a partial tile read before it is
    visible, or a ticket left non-zero, shows here), and agree with the same schedule without split K to the summation-order level."""
    from smap_amd.engine import BackboneEngine
    from smap_amd.model.smap import SMAP
    for k in ("SMAP_CAT", "SMAP_SKIPSUM", "SMAP_TAPHEAD"):          # (conftest forces round 6's launches on for the small TEST schedules; this is the real batch-1 schedule)
        monkeypatch.delenv(k, raising=False)
    torch.manual_seed(0)
    sd = recipe_state_dict(SMAP(make_cfg((128, 208))).state_dict())
    x = torch.randn(1, 3, 512, 832, generator=torch.Generator().manual_seed(7)).to(DEV)
    eng = BackboneEngine(sd, 1, 512, 832, DEV, precision="x3")
    assert sum(1 for op in eng.graph.ops if op.p.get("ksplit", 1) > 1) >= 30 and not any("cat" in op.p or "tap" in op.p for op in eng.graph.ops)
    sib = eng.sibling()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [eng.new_output(), sib.new_output()]
    ref = eng.new_output()
    eng.run(x, out=ref)
    torch.cuda.synchronize()
    ref_bits = ref.view(torch.int32).clone()
    bad = 0
    for it in range(150):
        for k, (e, st) in enumerate(((eng, streams[0]), (sib, streams[1]))):      # two executors of the plan in flight: shared tickets? no --
            with torch.cuda.stream(st):                                           # each has its own arena, hence its own ticket region
                e.run(x, out=outs[k])
        torch.cuda.synchronize()
        bad += int(not torch.equal(outs[0].view(torch.int32), ref_bits)) + int(not torch.equal(outs[1].view(torch.int32), ref_bits))
    assert bad == 0, bad
    monkeypatch.setenv("SMAP_SPLITK", "0")
    plain = BackboneEngine(sd, 1, 512, 832, DEV, precision="x3")
    assert not any(op.p.get("ksplit", 1) > 1 for op in plain.graph.ops)
    o = plain.new_output()
    plain.run(x, out=o)
    torch.cuda.synchronize()
    n = eng.out_floats
    assert (o[:n] - ref[:n]).abs().max().item() <= 2e-5 * ref[:n].abs().max().item()


SEG_CASES = [   # B, H, W, Cin, couts, relus, up (low-res size or None), tile
    (2, 13, 21, 256, (256, 64), (1, 1), None, 20),
    (2, 13, 21, 256, (256, 64, 256), (1, 1, 0), None, 21),
    (1, 16, 26, 512, (256, 512), (1, 1), (8, 13), 54),
    (2, 9, 11, 128, (128, 256, 8), (0, 1, 1), (5, 6), 2),
    (3, 16, 24, 256, (512, 256), (1, 0), None, 53),
    (1, 32, 52, 64, (256, 64), (1, 1), (16, 26), 50),
    (3, 16, 24, 256, (512, 256, 264), (1, 0, 1), None, 56),     # register-epilogue tile: every segment on its own 256-row N tile(s)
]


@pytest.mark.parametrize("x3", [True, False], ids=["x3", "f16"])
@pytest.mark.parametrize("case", SEG_CASES, ids=lambda c: "x".join(map(str, c[:4])) + "-" + "+".join(map(str, c[4])) + f"-t{c[7]}")
def test_merged_1x1_launch_matches_torch(case, x3):
    """One launch, several 1x1 convs on one input, one dense output tensor each (include/smap_hip.h smap_op.seg_*): every output
    against an f64 torch conv of the same (rounded) operands; ragged M, a segment narrower than the N tile, the fused bilinear
    add on segment 0 only, per-segment ReLU and accumulator scale."""
    import torch.nn.functional as F
    from smap_amd import engine as E
    from smap_amd import lib as L
    B, H, W, cin, couts, relus, up, tile = case
    if tile in E.REGEPI_TILES and not x3:
        pytest.skip("the register-epilogue tile has a split-precision instance only")
    gen = torch.Generator().manual_seed(sum(case[:4]) + tile)
    sd, segs = {}, []
    for j, (c, r) in enumerate(zip(couts, relus)):
        pre = f"s{j}"
        sd[pre + ".conv.weight"] = torch.randn(c, cin, 1, 1, generator=gen) * (0.05 * 4 ** j)      # different scales per segment
        sd[pre + ".conv.bias"] = torch.randn(c, generator=gen) * 0.1
        sd[pre + ".bn.weight"] = torch.rand(c, generator=gen) + 0.5
        sd[pre + ".bn.bias"] = torch.randn(c, generator=gen) * 0.1
        sd[pre + ".bn.running_mean"] = torch.randn(c, generator=gen) * 0.1
        sd[pre + ".bn.running_var"] = torch.rand(c, generator=gen) + 0.5
        segs.append((f"y{j}", pre, bool(r)))
    g = E.Graph(sd, B, H * 4, W * 4, keep_ref=True, precision="x3" if x3 else "f16", build=False)
    g.w_pairs = 1
    xt = g.tensor("x", H, W, cin)
    ut = g.tensor("up", up[0], up[1], couts[0]) if up else None
    outs = g.conv_seg(segs, xt, up=ut, tile=tile)
    xt.first = 0
    if ut is not None:
        ut.first = 0
    g.allocate(reuse=False)
    ops = g.emit()
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.smap_plan_create(ops, 1, C.byref(h)), "smap_plan_create")
    arena = torch.zeros(g.arena_bytes, dtype=torch.uint8, device=DEV)

    def put(t, v):                    # NCHW fp32 -> the tensor's storage (fp16, or hi | lo planes)
        v = v.permute(0, 2, 3, 1).contiguous()
        hi = v.to(torch.float16)
        if t.planes == 2:
            lo = (v - hi.float()).to(torch.float16)
            raw = torch.stack([hi, lo], 3).reshape(-1)
            val = (hi.double() + lo.double())
        else:
            raw, val = hi.reshape(-1), hi.double()
        arena[t.off:t.off + t.nbytes].view(torch.float16).copy_(raw.to(DEV))
        return val.permute(0, 3, 1, 2)
    xv = put(xt, torch.randn(B, cin, H, W, generator=gen))
    uv = put(ut, torch.randn(B, couts[0], up[0], up[1], generator=gen)) if up else None
    blob = g.weight_blob().to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    p = g.ops[0].p
    refs = [(p["w_ref"], p["b_ref"], p["relu"])] + [(sg["w_ref"], sg["b_ref"], sg["relu"]) for sg in p["segs"]]
    for j, (t, (w, b, r)) in enumerate(zip(outs, refs)):
        raw = arena[t.off:t.off + t.nbytes].view(torch.float16).cpu()
        got = (raw.view(B, H, W, 2, t.C).double().sum(3) if t.planes == 2 else raw.view(B, H, W, t.C).double()).permute(0, 3, 1, 2)
        wq = w.double() if x3 else w.to(torch.float16).double()
        want = F.conv2d(xv, wq, b.double())
        if j == 0 and up:
            want = want + F.interpolate(uv, size=(H, W), mode="bilinear", align_corners=True)
        if r:
            want = F.relu(want)
        err, mx = (got - want).abs().max().item(), want.abs().max().item()
        assert torch.isfinite(got).all() and err < ((3e-6 * mx + 1e-6) if x3 else 2e-3 * mx), (j, err, mx)


@pytest.mark.parametrize("x3", [True, False], ids=["x3", "f16"])
@pytest.mark.parametrize("shape", [(2, 16, 24), (1, 13, 21), (3, 5, 7), (1, 32, 52)], ids=lambda c: "x".join(map(str, c)))
def test_one_channel_3x3_head_as_tap_dots_and_a_stencil(shape, x3):
    """Graph.conv_tapdot + Graph.tapsum (smap_op.tap_n, SMAP_OP_TAPSUM): conv3x3_{256->1}(relu(conv1x1_{256->256}(x))) -- the root-depth head,
    smap.py:227-229 -- with the 256-channel activation never stored: the 1x1 launch keeps nine dot products per pixel, a stencil sums them.
    Against the f64 evaluation of the two convs; ragged M tiles, maps smaller than a tile, zero padding at the border."""
    import torch.nn.functional as F
    from smap_amd import engine as E
    from smap_amd import lib as L
    B, H, W = shape
    gen = torch.Generator().manual_seed(B * 100 + H)
    sd = {}
    for pre, co, ci, k in (("c1", 256, 256, 1), ("c3", 1, 256, 3)):
        sd[pre + ".conv.weight"] = torch.randn(co, ci, k, k, generator=gen) * (1.0 / (ci * k * k)) ** 0.5
        sd[pre + ".conv.bias"] = torch.randn(co, generator=gen) * 0.1
        sd[pre + ".bn.weight"] = torch.rand(co, generator=gen) + 0.5
        sd[pre + ".bn.bias"] = torch.randn(co, generator=gen) * 0.1
        sd[pre + ".bn.running_mean"] = torch.randn(co, generator=gen) * 0.1
        sd[pre + ".bn.running_var"] = torch.rand(co, generator=gen) + 0.5
    g = E.Graph(sd, B, 4 * H, 4 * W, keep_ref=True, precision="x3" if x3 else "f16", build=False)
    g.w_pairs = int(B == 1)
    g.out_h, g.out_w, g.status_off, g.status_words = H, W, B * H * W * 4, 1
    xt = g.tensor("x", H, W, 256)
    t, b3 = g.conv_tapdot("t", "c1", "c3", xt)
    g.tapsum(t, b3, 0)
    xt.first = 0
    g.allocate(reuse=False)
    ops = g.emit()
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.smap_plan_create(ops, 2, C.byref(h)), "smap_plan_create")
    arena = torch.zeros(g.arena_bytes, dtype=torch.uint8, device=DEV)
    v = torch.randn(B, 256, H, W, generator=gen).permute(0, 2, 3, 1).contiguous()
    hi = v.to(torch.float16)
    if xt.planes == 2:
        lo = (v - hi.float()).to(torch.float16)
        raw, val = torch.stack([hi, lo], 3).reshape(-1), hi.double() + lo.double()
    else:
        raw, val = hi.reshape(-1), hi.double()
    arena[xt.off:xt.off + xt.nbytes].view(torch.float16).copy_(raw.to(DEV))
    xv = val.permute(0, 3, 1, 2)
    blob = g.weight_blob().to(DEV)
    out = torch.full((B * H * W + 1,), 7.0, dtype=torch.float32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), C.c_void_p(out.data_ptr()), st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    p = g.ops[0].p
    q = (lambda w: w.double()) if x3 else (lambda w: w.to(torch.float16).double())
    y = F.relu(F.conv2d(xv, q(p["w_ref"]), p["b_ref"].double()))
    want = F.conv2d(y, p["tap"]["w_ref"].double(), g.ops[1].p["b_ref"].double(), padding=1)
    got = out[:B * H * W].view(B, 1, H, W).cpu().double()
    err, mx = (got - want).abs().max().item(), want.abs().max().item()
    assert int(out[-1:].view(torch.int32).item()) == 0 and err < ((3e-6 * mx + 1e-6) if x3 else 2e-3 * mx), (err, mx)
    tt = arena[t.off:t.off + t.nbytes].view(torch.float32).view(B, H, W, 16).cpu()
    assert not tt[..., 9:].any()                                                 # the padding lanes of a pixel's 16 floats
    bad = (L.SmapOp * 2)(ops[0], ops[1])
    bad[0].tile = 50                                                             # only the one-N-tile 128 x 256 instance has the epilogue
    assert lib.smap_plan_create(bad, 2, C.byref(h)) == -1


CAT_CASES = [   # B, H (out), W (out), planes (c3's K), in_planes (the shortcut's K), cout, spatial stride of the shortcut, tile
    (2, 8, 13, 128, 256, 512, 2, 50),          # layer2's first block in small: ragged M tile, stride 2 (odd input width 25 -> 13)
    (1, 16, 26, 256, 512, 1024, 2, 51),        # layer3's
    (3, 5, 7, 512, 1024, 2048, 2, 20),         # layer4's: M < one tile
    (2, 16, 24, 64, 64, 256, 1, 50),           # layer1's first block (stride 1: the shortcut reads the block's own input)
    (1, 9, 11, 128, 192, 320, 1, 20),          # Cout not a tile multiple
    (2, 8, 13, 128, 256, 512, 2, 54),          # 128 x 256 and 256 x 128 instances
    (1, 16, 26, 256, 512, 1024, 2, 53),
]


@pytest.mark.parametrize("x3", [True, False], ids=["x3", "f16"])
@pytest.mark.parametrize("case", CAT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_last_1x1_with_the_shortcut_conv_as_one_gemm(case, x3):
    """Graph.conv_cat / smap_op.in2_*: relu(W3 y + Wd x[::s, ::s] + b3 + bd) -- a Bottleneck's last 1x1 and its 1x1 shortcut conv
    (smap.py:60-77, 124-129) as ONE launch over K = (planes | in_planes) -- against the f64 evaluation of the two convs on the same (rounded)
    operands."""
    import torch.nn.functional as F
    from smap_amd import engine as E
    from smap_amd import lib as L
    B, H, W, c1, c2, cout, st2, tile = case
    gen = torch.Generator().manual_seed(sum(case))
    sd = {}
    for pre, c, sc in (("c3", c1, 0.06), ("ds", c2, 0.21)):           # different weight scales: ONE power-of-two scale serves both halves of K
        sd[pre + ".conv.weight"] = torch.randn(cout, c, 1, 1, generator=gen) * sc
        sd[pre + ".conv.bias"] = torch.randn(cout, generator=gen) * 0.1
        sd[pre + ".bn.weight"] = torch.rand(cout, generator=gen) + 0.5
        sd[pre + ".bn.bias"] = torch.randn(cout, generator=gen) * 0.1
        sd[pre + ".bn.running_mean"] = torch.randn(cout, generator=gen) * 0.1
        sd[pre + ".bn.running_var"] = torch.rand(cout, generator=gen) + 0.5
    H2, W2 = (H - 1) * st2 + 1 + (st2 - 1) * (W % 2), (W - 1) * st2 + 1          # (H2 - 1) // s + 1 == H either way
    g = E.Graph(sd, B, 4 * H, 4 * W, keep_ref=True, precision="x3" if x3 else "f16", build=False)
    g.w_pairs = int(B == 1)
    yt, xt = g.tensor("y", H, W, c1), g.tensor("x", H2, W2, c2)
    out = g.conv_cat("out", "c3", yt, "ds", xt, st2, relu=True, tile=tile)
    yt.first = xt.first = 0
    g.allocate(reuse=False)
    ops = g.emit()
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.smap_plan_create(ops, 1, C.byref(h)), "smap_plan_create")
    arena = torch.zeros(g.arena_bytes, dtype=torch.uint8, device=DEV)

    def put(t, v):
        v = v.permute(0, 2, 3, 1).contiguous()
        hi = v.to(torch.float16)
        if t.planes == 2:
            lo = (v - hi.float()).to(torch.float16)
            raw, val = torch.stack([hi, lo], 3).reshape(-1), hi.double() + lo.double()
        else:
            raw, val = hi.reshape(-1), hi.double()
        arena[t.off:t.off + t.nbytes].view(torch.float16).copy_(raw.to(DEV))
        return val.permute(0, 3, 1, 2)
    yv = put(yt, torch.randn(B, c1, H, W, generator=gen))
    xv = put(xt, torch.randn(B, c2, H2, W2, generator=gen))
    blob = g.weight_blob().to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    p = g.ops[0].p
    q = (lambda w: w.double()) if x3 else (lambda w: w.to(torch.float16).double())
    want = F.relu(F.conv2d(yv, q(p["w_ref"]), p["b_ref"].double()) + F.conv2d(xv[:, :, ::st2, ::st2], q(p["cat"]["w_ref"]), p["cat"]["b_ref"].double()))
    raw = arena[out.off:out.off + out.nbytes].view(torch.float16).cpu()
    got = (raw.view(B, H, W, 2, out.C).double().sum(3) if out.planes == 2 else raw.view(B, H, W, out.C).double()).permute(0, 3, 1, 2)
    err, mx = (got - want).abs().max().item(), want.abs().max().item()
    assert torch.isfinite(got).all() and err < ((3e-6 * mx + 1e-6) if x3 else 2e-3 * mx), (err, mx)
    # and the plan refuses what the kernel cannot do: a second input on a tile without the instance, with split K, beyond the first input's window
    bad = (L.SmapOp * 1)(ops[0])
    bad[0].tile = 21
    assert lib.smap_plan_create(bad, 1, C.byref(h)) == -1
    bad = (L.SmapOp * 1)(ops[0])
    bad[0].in2_off = ops[0].in2_off + (1 << 32)
    assert lib.smap_plan_create(bad, 1, C.byref(h)) == -1
    bad = (L.SmapOp * 1)(ops[0])
    bad[0].in2_stride = 3
    assert lib.smap_plan_create(bad, 1, C.byref(h)) == -1


@pytest.mark.parametrize("x3", [True, False], ids=["x3", "f16"])
@pytest.mark.parametrize("case", [(2, 8, 13, 512, 256, 512, 50), (1, 16, 26, 1024, 256, 1024, 51), (3, 5, 7, 2048, 256, 2048, 50), (2, 16, 24, 256, 256, 256, 51),
                                  (1, 9, 11, 128, 192, 320, 50), (2, 8, 13, 512, 256, 512, 54), (1, 16, 26, 1024, 256, 1024, 53)], ids=lambda c: "x".join(map(str, c)))
def test_two_activated_skip_convs_as_one_launch_and_one_tensor(case, x3):
    """Graph.conv_relusum / smap_op.in2_mode = 1: relu(skip1(x)) + relu(skip2(out)) (smap.py:218-241; only their sum is ever used, :142-153) as ONE
    launch writing ONE tensor -- the first conv's accumulators are activated and parked in registers while the second conv runs -- against the
    f64 evaluation of the two convs on the same operands; each conv with its own power-of-two weight scale (x 40 apart here)."""
    import torch.nn.functional as F
    from smap_amd import engine as E
    from smap_amd import lib as L
    B, H, W, c1, c2, cout, tile = case
    gen = torch.Generator().manual_seed(sum(case))
    sd = {}
    for pre, c, sc in (("s1", c1, 0.004), ("s2", c2, 0.16)):
        sd[pre + ".conv.weight"] = torch.randn(cout, c, 1, 1, generator=gen) * sc
        sd[pre + ".conv.bias"] = torch.randn(cout, generator=gen) * 0.1
        sd[pre + ".bn.weight"] = torch.rand(cout, generator=gen) + 0.5
        sd[pre + ".bn.bias"] = torch.randn(cout, generator=gen) * 0.1
        sd[pre + ".bn.running_mean"] = torch.randn(cout, generator=gen) * 0.1
        sd[pre + ".bn.running_var"] = torch.rand(cout, generator=gen) + 0.5
    g = E.Graph(sd, B, 4 * H, 4 * W, keep_ref=True, precision="x3" if x3 else "f16", build=False)
    g.w_pairs = int(B == 1)
    xt, ot = g.tensor("x", H, W, c1), g.tensor("o", H, W, c2)
    out = g.conv_relusum("s", "s1", xt, "s2", ot, tile=tile)
    xt.first = ot.first = 0
    g.allocate(reuse=False)
    ops = g.emit()
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.smap_plan_create(ops, 1, C.byref(h)), "smap_plan_create")
    arena = torch.zeros(g.arena_bytes, dtype=torch.uint8, device=DEV)

    def put(t, v):
        v = v.permute(0, 2, 3, 1).contiguous()
        hi = v.to(torch.float16)
        if t.planes == 2:
            lo = (v - hi.float()).to(torch.float16)
            raw, val = torch.stack([hi, lo], 3).reshape(-1), hi.double() + lo.double()
        else:
            raw, val = hi.reshape(-1), hi.double()
        arena[t.off:t.off + t.nbytes].view(torch.float16).copy_(raw.to(DEV))
        return val.permute(0, 3, 1, 2)
    xv, ov = put(xt, torch.randn(B, c1, H, W, generator=gen) * 3), put(ot, torch.randn(B, c2, H, W, generator=gen))
    blob = g.weight_blob().to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.smap_plan_run(h, None, C.c_void_p(arena.data_ptr()), C.c_void_p(blob.data_ptr()), None, st), "run")
    torch.cuda.synchronize()
    lib.smap_plan_destroy(h)
    p = g.ops[0].p
    q = (lambda w: w.double()) if x3 else (lambda w: w.to(torch.float16).double())
    a, b = F.conv2d(xv, q(p["w_ref"]), p["b_ref"].double()), F.conv2d(ov, q(p["cat"]["w_ref"]), p["cat"]["b_ref"].double())
    want = F.relu(a) + F.relu(b)
    assert (a < 0).any() and (b < 0).any() and (a > 0).any() and (b > 0).any()          # both activations matter
    raw = arena[out.off:out.off + out.nbytes].view(torch.float16).cpu()
    got = (raw.view(B, H, W, 2, out.C).double().sum(3) if out.planes == 2 else raw.view(B, H, W, out.C).double()).permute(0, 3, 1, 2)
    err, mx = (got - want).abs().max().item(), want.abs().max().item()
    assert torch.isfinite(got).all() and err < ((3e-6 * mx + 1e-6) if x3 else 2e-3 * mx), (err, mx)
    bad = (L.SmapOp * 1)(ops[0])
    bad[0].tile = 20                                                                   # K-concatenation yes, relu-sum no (288 registers)
    assert lib.smap_plan_create(bad, 1, C.byref(h)) == -1
    bad = (L.SmapOp * 1)(ops[0])
    bad[0].relu = 1
    assert lib.smap_plan_create(bad, 1, C.byref(h)) == -1


@pytest.mark.parametrize("env", [{"SMAP_HALO3": "16"}, {"SMAP_HALO3": "32", "SMAP_HALO3_DEEP": "1"}], ids=["halo16", "halo32deep"])
def test_small_schedule_split_precision_with_halo_kernel(golden_dir, small, monkeypatch, env):
    """precision "x3" with every plain 3x3 conv on the halo-tiled kernel's split-precision instances."""
    from smap_amd.engine import BackboneEngine
    _, sd = small
    z = np.load(f"{golden_dir}/backbone_small.npz")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False, precision="x3")
    assert any(30 <= op.p["tile"] < 50 for op in eng.graph.ops if op.kind == 0)
    outs = [o.cpu() for o in eng.run(torch.from_numpy(z["x"]).to(DEV))]
    for a, k in zip(outs, ("hms", "det_d", "root_d")):
        assert np.abs(a.numpy() - z[k]).max() < 2e-5 * np.abs(z[k]).max(), k


@pytest.mark.parametrize("precision,tol", [("x3", 2e-5), ("f16", 1e-2)])
@pytest.mark.parametrize("spec", ["64:80", "64:81,128:82"])
def test_small_schedule_with_fused_bottleneck_tails(golden_dir, small, monkeypatch, spec, precision, tol):
    """Stride-1 Bottlenecks with c2 + c3 as one launch (csrc/convf.hip): every stored tensor against the f64 interpretation of
    the SAME schedule, and the outputs against the golden outputs of the reference model."""
    from smap_amd.engine import BackboneEngine, Graph
    from oracle.graph_interp import run_graph
    _, sd = small
    z = np.load(f"{golden_dir}/backbone_small.npz")
    x = torch.from_numpy(z["x"])
    monkeypatch.setenv("SMAP_TAIL", spec)
    monkeypatch.setenv("SMAP_BLOCK", "")                 # (the whole-block launches would take layer1's blocks in split precision)
    monkeypatch.setenv("SMAP_BLOCK_FIRST", "")
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False, precision=precision)
    n_tail = sum(1 for op in eng.graph.ops if op.kind == 0 and "tail" in op.p)
    assert n_tail == 3 * (3 if spec == "64:80" else 3 + 3) and not any(t.name.endswith(".c2") for t in eng.graph.tensors if ".layer1.1" in t.name)
    outs = [o.cpu() for o in eng.run(x.to(DEV))]
    torch.cuda.synchronize()
    g = Graph(sd, 2, 64, 96, keep_ref=True)
    with torch.no_grad():
        *ref, T = run_graph(g, x.double(), quantize=False, keep=True)
    worst = []
    for t in eng.graph.tensors:
        got = eng.read_tensor(t.name).cpu().double().permute(0, 3, 1, 2)
        want = T[t.name].double()
        c = want.shape[1]
        worst.append(((got[:, :c] - want).abs().max().item() / (want.abs().max().item() + 1e-6), t.name))
    worst.sort(reverse=True)
    assert worst[0][0] < tol, worst[:5]
    for a, k in zip(outs, ("hms", "det_d", "root_d")):
        assert np.abs(a.numpy() - z[k]).max() < tol * np.abs(z[k]).max(), k


@pytest.mark.parametrize("env", [{"SMAP_HALO3": "16"}, {"SMAP_HALO3": "32", "SMAP_HALO3_DEEP": "1"}], ids=["halo16", "halo32deep"])
def test_small_schedule_with_specialised_kernels(golden_dir, small, monkeypatch, env):
    """The whole schedule with every eligible 3x3 conv routed to the halo-tiled kernel (fp32 heads included) against the
    golden outputs."""
    from smap_amd.engine import BackboneEngine
    _, sd = small
    z = np.load(f"{golden_dir}/backbone_small.npz")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = BackboneEngine(sd, 2, 64, 96, DEV, reuse=False)
    tiles = {op.p["tile"] for op in eng.graph.ops if op.kind == 0}
    assert any(t >= 30 for t in tiles), tiles
    outs = [o.cpu() for o in eng.run(torch.from_numpy(z["x"]).to(DEV))]
    for a, k in zip(outs, ("hms", "det_d", "root_d")):
        assert np.abs(a.numpy() - z[k]).max() < 1e-2 * np.abs(z[k]).max(), k


@pytest.mark.parametrize("precision", ["f16", "x3"])
def test_flip_tta_inside_the_schedule_is_bit_exact(small, precision):
    """Engine built with flip_pair: the stem reads the mirrored image by index, the head sum merges the mirrored maps
    (test.py:55-70).  Must equal, bit for bit, two separate forwards (frames, torch.flip(frames)) merged by the reference's
    channel loop; det_d / root_d are those of the un-mirrored pass."""
    from exps.stage3_root2.config import cfg
    from exps.stage3_root2.test_util import merge_flip
    net, sd = small
    net = net.to(DEV)
    net.precision = precision
    kpt = cfg.DATASET.KEYPOINT.NUM
    pair = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [kpt + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
    x = torch.randn(3, 3, 64, 96, generator=torch.Generator().manual_seed(21)).to(DEV)
    h, d, rd = net(x)
    hf, _, _ = net(torch.flip(x, [-1]))
    want = merge_flip(h.clone(), hf, cfg)
    eng = net.engine(3, 64, 96, torch.device(DEV), flip_pair=pair)
    assert eng.graph.B == 6 and eng.B == 3
    got_h, got_d, got_rd = eng.run(x)
    torch.cuda.synchronize()
    assert got_h.shape == want.shape and torch.equal(got_h, want)
    assert torch.equal(got_d, d) and torch.equal(got_rd, rd)
    assert (want - h).abs().max().item() > 1e-3            # the merge did something
    net.precision = "f16"


def test_smap_module_forward_and_reload(golden_dir, small):
    from model.smap import SMAP
    net, sd = small
    z = np.load(f"{golden_dir}/backbone_small.npz")
    net = net.to(DEV)
    hms, det_d, root_d = net(torch.from_numpy(z["x"]).to(DEV))
    assert hms.shape == (2, 43, 16, 24) and det_d.shape == (2, 14, 16, 24) and root_d.shape == (2, 1, 16, 24)
    assert hms.dtype == torch.float32 and hms.is_cuda
    assert np.abs(hms.cpu().numpy() - z["hms"]).max() < 1e-2 * np.abs(z["hms"]).max()
    # a strict load of a reference-keyed checkpoint rebuilds the engine with the new weights
    sd2 = {k: (v * 0.5 if k.endswith("res_conv2.bn.weight") else v) for k, v in sd.items()}
    net.load_state_dict(sd2, strict=True)
    hms2, _, _ = net(torch.from_numpy(z["x"]).to(DEV))
    assert (hms2 - hms).abs().max().item() > 1e-3


def test_full_size_properties():
    """3x512x832 (BASELINE config): determinism, batch independence, flip sanity, finiteness."""
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = recipe_state_dict(net.state_dict())
    net.load_state_dict(sd)
    net = net.to(DEV)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(3, 3, 512, 832, generator=g).to(DEV)
    a = [t.clone() for t in net(x)]
    b = [t.clone() for t in net(x)]
    for u, v in zip(a, b):
        assert torch.equal(u, v) and torch.isfinite(u).all()
    assert a[0].shape == (3, 43, 128, 208) and a[1].shape == (3, 14, 128, 208) and a[2].shape == (3, 1, 128, 208)
    perm = torch.tensor([2, 0, 1], device=DEV)
    c = net(x[perm])
    for u, v in zip(a, c):
        assert torch.equal(u[perm], v)                     # frames are independent (eval-mode BN)
    single = net(x[1:2])
    for u, v in zip(a, single):                            # different batch -> different tiling of M
        assert (u[1:2] - v).abs().max().item() <= 2e-2 * u.abs().max().item()
    assert a[0].abs().max().item() > 1e-3


@pytest.mark.parametrize("precision,tol_max,tol_mean", [("x3", 1e-4, 2e-5), ("f16", 1e-2, 2e-3)])
def test_full_size_forward_vs_cpu_oracle(precision, tol_max, tol_mean):
    """BASELINE configs[1]: batch=1, 3x512x832, forward only -- HIP engine vs the CPU restatement of the
    reference forward (oracle/backbone_ref.py, fp32), recipe weights.  Split precision (the default) is held to the
    fp32-roundoff level SURVEY.md 7 step 4 asks of an fp32-equivalent path; plain fp16 to its storage tolerance."""
    from smap_amd.model.smap import SMAP
    from oracle.backbone_ref import smap_forward
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = recipe_state_dict(net.state_dict())
    net.load_state_dict(sd)
    x = torch.randn(1, 3, 512, 832, generator=torch.Generator().manual_seed(1234))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = smap_forward(sd, x)
    net.precision = precision
    out = [t.cpu() for t in net.to(DEV)(x.to(DEV))]
    for a, b, k in zip(out, ref, ("hms", "det_d", "root_d")):
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err < tol_max, (k, err)
        # the bulk of the map is far tighter than the worst pixel
        assert ((a - b).abs().mean() / b.abs().mean()).item() < tol_mean, k


def test_full_size_forward_vs_imported_reference_digest(golden_dir):
    """The HIP split-precision path at BASELINE configs[1] against the IMPORTED reference model itself (not the restatement):
    tests/golden/backbone_full.npz holds per-channel sums and 1024 sampled positions of each output of model.smap.SMAP at
    1x3x512x832 (tests/golden/gen_golden_full.py); 1e-4 as for the oracle comparison above."""
    from smap_amd.model.smap import SMAP
    from test_oracle_cpu import check_against_full_size_digest, full_size_input
    z = np.load(f"{golden_dir}/backbone_full.npz")
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    net.load_state_dict(recipe_state_dict(net.state_dict()))
    net.precision = "x3"
    x = full_size_input(z)
    outs = [t.cpu() for t in net.to(DEV)(x.to(DEV))]
    worst = check_against_full_size_digest(z, outs, 1e-4)
    print("HIP x3 vs imported reference at full size (sampled, channel sums):", worst)


@pytest.mark.parametrize("scale,stem_gain", [(1e-4, 1.0), (1.0, 1.0), (3e3, 1.0), (1e5, 1.0), (1.0, 1e-4), (1.0, 1e3), (1.0, 1e5)])
def test_split_precision_dynamic_range(scale, stem_gain):
    """Split precision keeps fp16's RANGE: values around 1e-4 have subnormal lo parts, values beyond 65504 turn into inf.
    `scale` multiplies the input image; `stem_gain` multiplies the folded output of the stem (BN weight and bias of top.conv),
    i.e. the activations that enter layer1 of every stage-0 block (ReLU and max-pool are homogeneous).  The contract is EITHER
    maps within 1e-4 of the fp32 reference OR the status word set (RuntimeError from raise_if_nonfinite / PosePipeline) --
    never silently wrong maps."""
    from smap_amd.model.smap import SMAP
    from oracle.backbone_ref import smap_forward
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()                  # (its own module: other tests edit the shared fixture's weights)
    sd = recipe_state_dict(net.state_dict())
    sd["top.conv.bn.weight"] = sd["top.conv.bn.weight"] * stem_gain
    sd["top.conv.bn.bias"] = sd["top.conv.bn.bias"] * stem_gain
    net.load_state_dict(sd)
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(4)) * scale
    with torch.no_grad():
        ref = smap_forward(sd, x)
    net.precision = "x3"
    eng = net.to(DEV).engine(1, 64, 96, torch.device(DEV))
    out = eng.new_output()
    got = [t.cpu() for t in eng.run(x.to(DEV), out=out)]
    finite = all(torch.isfinite(t).all() for t in got)
    assert (eng.status(out) & 1) == (0 if finite else 1)                # the kernel-side guard sees exactly what isfinite sees
    if not finite:
        with pytest.raises(RuntimeError, match="fp16 range"):
            eng.raise_if_nonfinite(out)
        assert scale >= 3e3 or stem_gain >= 1e3, "only large activations may overflow"
        return
    for a, b, k in zip(got, ref, ("hms", "det_d", "root_d")):
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err < 1e-4, (k, scale, stem_gain, err)


@pytest.mark.parametrize("B,flip,precision", [(2, True, "x3"), (8, False, "x3"), (1, True, "f16")])
def test_overlapping_executors_are_deterministic(B, flip, precision):
    """Two executors of one schedule on two streams, many launches queued back to back (what PosePipeline(depth=2) does at
    512x832): every output must equal the serial result bit for bit.  Until round 3 the head-sum kernel kept a run-time-indexed
    private array; the compiler promoted it to LDS (a per-thread slot addressed through the dispatch packet) and 10-30 % of
    OVERLAPPED forwards came back with ~15 wrong map values (never a serial one: every other test in this file is serial)."""
    from exps.stage3_root2.config import cfg
    from smap_amd.engine import BackboneEngine
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    sd = recipe_state_dict(SMAP(make_cfg((128, 208))).state_dict())
    fp = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [cfg.DATASET.KEYPOINT.NUM + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
    eng = BackboneEngine(sd, B, 512, 832, DEV, precision=precision, flip_pair=fp if flip else None)
    engs = [eng, eng.sibling()]
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    imgs = torch.randn(B, 3, 512, 832, generator=torch.Generator().manual_seed(3)).to(DEV)
    ref = eng.new_output()
    eng.run(imgs, out=ref)
    torch.cuda.synchronize()
    N = 24
    outs = [eng.new_output() for _ in range(N)]
    torch.cuda.synchronize()
    for i in range(N):
        with torch.cuda.stream(streams[i % 2]):
            engs[i % 2].run(imgs, out=outs[i])
    torch.cuda.synchronize()
    bad = [i for i in range(N) if not torch.equal(outs[i], ref)]
    assert not bad, bad


def test_graph_replay_equals_direct_launches(small):
    """BackboneEngine.capture: the schedule replayed as one HIP graph writes the same bytes as the launch-by-launch run,
    for fresh inputs and repeated replays."""
    from smap_amd.engine import BackboneEngine
    _, sd = small
    eng = BackboneEngine(sd, 2, 64, 96, DEV)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(2, 3, 64, 96, generator=g).to(DEV) for _ in range(3)]
    want = [[t.clone() for t in eng.run(x)] for x in xs]
    out = eng.new_output()
    replay = eng.capture(out)
    for rep in range(2):
        for x, w in zip(xs, want):
            got = replay(x)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(got, w))


@pytest.mark.parametrize("precision,flip", [("f16", False), ("x3", False), ("x3", True)])
@pytest.mark.parametrize("hw", [(64, 96), (96, 160), (512, 832)])
def test_stem_pool_fusion_is_bit_exact(small, monkeypatch, hw, precision, flip):
    """ResNet_top as one kernel (stem_pool_kernel: conv tile in LDS, 3x3 s2 max from there) writes the same pooled tensor
    as stem_kernel + maxpool_kernel, bit for bit, including ragged pooled tiles and image borders -- in both precisions and
    with the mirrored half of a flip-TTA batch."""
    from smap_amd.engine import BackboneEngine, OP_STEMPOOL
    _, sd = small
    H, W = hw
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(H)).to(DEV) * 2
    fp = list(range(43)) if flip else None
    monkeypatch.setenv("SMAP_STEMPOOL", "1")
    eng = BackboneEngine(sd, 1, H, W, DEV, reuse=False, precision=precision, flip_pair=fp)
    assert eng.graph.ops[0].kind == OP_STEMPOOL
    eng.run(x, first=0, count=1)
    got = eng.read_tensor("top.pool").clone()
    monkeypatch.delenv("SMAP_STEMPOOL")
    ref = BackboneEngine(sd, 1, H, W, DEV, reuse=False, precision=precision, flip_pair=fp)
    ref.run(x, first=0, count=2)
    want = ref.read_tensor("top.pool")
    torch.cuda.synchronize()
    assert got.shape == want.shape == (2 if flip else 1, H // 4, W // 4, 64)
    assert torch.equal(got, want)
    assert want.abs().max() > 0
