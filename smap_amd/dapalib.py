"""Host-side mirror of the reference's `dapalib` extension module.

Reference: extensions/association.cpp:236-241 exports
    dapalib.connect(hmsIn, rDepth, rootIdx=2, distFlag=True) -> Tensor
    dapalib.extract(hmsIn) -> (list[Tensor] x 15, list[Tensor] x 14)
with hmsIn a contiguous fp32 CUDA tensor [43,128,208] (association.cpp:21,51) and
rDepth a [128,208] fp32 tensor indexed on the host (association.cpp:140).  Same
names, argument meaning, defaults and return types here; the work is done by the
HIP kernels of libsmap_hip.so (include/smap_hip.h) on the current PyTorch stream.

Differences, all additive:
  * arguments are validated (the reference has no checks: wrong shape/device = UB)
    and errors raise ValueError / smap_amd.lib.SmapError;
  * any H x W with H*W <= 32768 is accepted, not only 128 x 208;
  * `connect_batch`, `extract_batch`, `lift_batch`, `refine_batch` process B frames
    per launch and keep results on the device (the reference notes "no batch
    implementation yet", test.py:114).
"""
import ctypes as C

import torch

from . import lib as _L

NJ, NL, MAXP, HMS_C = 15, 14, 127, 43
JOINT_PAIRS = [0, 1, 0, 2, 0, 9, 9, 10, 10, 11, 0, 3, 3, 4, 4, 5, 2, 12, 12, 13, 13, 14, 2, 6, 6, 7, 7, 8]


def _stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _check_hms(hms, batched):
    nd = 4 if batched else 3
    if not isinstance(hms, torch.Tensor):
        raise ValueError("hmsIn must be a torch.Tensor")
    if hms.dim() != nd or hms.shape[-3] != HMS_C:
        raise ValueError(f"hmsIn must have shape {'[B,' if batched else '['}43,H,W], got {tuple(hms.shape)}")
    if hms.dtype != torch.float32:
        raise ValueError(f"hmsIn must be float32, got {hms.dtype}")
    if not hms.is_cuda:
        raise ValueError("hmsIn must live on the GPU (the reference reads it with a device-to-device copy)")
    if not hms.is_contiguous():
        raise ValueError("hmsIn must be contiguous")
    H, W = hms.shape[-2:]
    if H < 3 or W < 3 or H * W > 32768:
        raise ValueError(f"unsupported heat-map size {H}x{W}")
    return H, W


def scale_hms_(hms):
    """In place hms[...,:15,:,:] /= 255 ; hms[...,15:,:,:] /= 127 (test.py:111-112)."""
    batched = hms.dim() == 4
    H, W = _check_hms(hms, batched)
    B = hms.shape[0] if batched else 1
    with torch.cuda.device(hms.device):
        _L.check(_L.load().smap_scale_hms(_p(hms), B, H, W, _stream(hms.device)), "smap_scale_hms")
    return hms


def flip_merge_(hms, hms_flip, pair):
    """In place flip-TTA merge (test.py:55-70) of hms [B,43,H,W] with the heat-maps of the mirrored
    images; pair = KEYPOINT.FLIP_ORDER + [15 + c for c in PAF.FLIP_CHANNEL]."""
    H, W = _check_hms(hms, True)
    _check_hms(hms_flip, True)
    if hms_flip.shape != hms.shape or len(pair) != HMS_C:
        raise ValueError("hms_flip must match hms and pair must have 43 entries")
    tab = (C.c_int * HMS_C)(*[int(p) for p in pair])
    with torch.cuda.device(hms.device):
        _L.check(_L.load().smap_flip_merge(_p(hms), _p(hms_flip), tab, hms.shape[0], H, W, _stream()), "smap_flip_merge")
    return hms


def extract_batch(hms, fused_nms=False):
    """hms [B,43,H,W] -> (peaks [B,15,128,3], scores [B,14,127,127]) on the device.  The peak search runs as smap_nms_ws (mask of the
    whole batch with one thread per pixel, then scan + centroids per channel; workspace from torch's caching allocator); fused_nms=True:
    the single-launch smap_nms (same peaks bit for bit, tests/test_ref_gpu.py)."""
    H, W = _check_hms(hms, True)
    B = hms.shape[0]
    lib = _L.load()
    peaks = torch.empty((B, NJ, MAXP + 1, 3), dtype=torch.float32, device=hms.device)
    scores = torch.empty((B, NL, MAXP, MAXP), dtype=torch.float32, device=hms.device)
    with torch.cuda.device(hms.device):
        if fused_nms:
            _L.check(lib.smap_nms(_p(hms), B, HMS_C, H, W, 0.2, _p(peaks), _stream()), "smap_nms")
        else:
            nb = lib.smap_nms_workspace_bytes(B, H, W)
            ws = torch.empty((nb // 8,), dtype=torch.int64, device=hms.device)
            _L.check(lib.smap_nms_ws(_p(hms), B, HMS_C, H, W, 0.2, _p(peaks), _p(ws), nb, _stream()), "smap_nms_ws")
        _L.check(lib.smap_paf_score(_p(hms), _p(peaks), B, H, W, _p(scores), _stream()), "smap_paf_score")
    return peaks, scores


def connect_batch(hms, rdepth, rootIdx=2, distFlag=True, return_intermediate=False, fused_nms=False, counts=None):
    """Batched connect: hms [B,43,H,W], rdepth [B,H,W] (or [B,1,H,W]) ->
    (bodys [B,127,15,4] fp32, counts [B] int32), both on the device."""
    H, W = _check_hms(hms, True)
    B = hms.shape[0]
    if rdepth.dim() == 4 and rdepth.shape[1] == 1:
        rdepth = rdepth[:, 0]
    if tuple(rdepth.shape) != (B, H, W):
        raise ValueError(f"rDepth must have shape [B,{H},{W}], got {tuple(rdepth.shape)}")
    if not 0 <= int(rootIdx) < NJ:
        raise ValueError("rootIdx out of range")
    rdepth = rdepth.to(device=hms.device, dtype=torch.float32).contiguous()
    peaks, scores = extract_batch(hms, fused_nms=fused_nms)
    bodys = torch.empty((B, MAXP, NJ, 4), dtype=torch.float32, device=hms.device)
    if counts is None:
        counts = torch.empty((B,), dtype=torch.int32, device=hms.device)
    elif tuple(counts.shape) != (B,) or counts.dtype != torch.int32 or counts.device != hms.device or not counts.is_contiguous():
        raise ValueError("connect_batch(counts=...): a contiguous int32 [B] tensor on the maps' device")
    with torch.cuda.device(hms.device):
        _L.check(_L.load().smap_group(_p(peaks), _p(scores), _p(rdepth), B, H, W, int(rootIdx),
                                      int(bool(distFlag)), _p(bodys), _p(counts), _stream()), "smap_group")
    if return_intermediate:
        return bodys, counts, peaks, scores
    return bodys, counts


def register_gt_batch(bodys, counts, gt_roots, gt_counts):
    """register_pred WITH ground truth (test_util.py:18-42) for the generate_result / generate_train modes.
    bodys [B,127,15,4] / counts [B]: connect_batch's outputs; gt_roots [B,G,2] fp32: root joint (x,y), network
    pixels, of the annotations whose root is visible (test.py:76-80), zero padded; gt_counts [B] int.
    -> matched [B,127,15,4] fp32 (row g = prediction assigned to annotation g, or zeros), matched_counts [B]
    (= gt_counts, or 0 for frames the reference skips: no prediction or no kept annotation)."""
    B = bodys.shape[0]
    dev = bodys.device
    gt_roots = torch.as_tensor(gt_roots, dtype=torch.float32).to(dev).contiguous()
    gt_counts = torch.as_tensor(gt_counts).to(device=dev, dtype=torch.int32).contiguous()
    if gt_roots.dim() != 3 or gt_roots.shape[0] != B or gt_roots.shape[2] != 2 or not 0 < gt_roots.shape[1] <= 64:
        raise ValueError("gt_roots must be [B,G,2] with 1 <= G <= 64")
    if tuple(gt_counts.shape) != (B,):
        raise ValueError("gt_counts must be [B]")
    matched = torch.empty((B, MAXP, NJ, 4), dtype=torch.float32, device=dev)
    mcounts = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _L.check(_L.load().smap_register_gt(_p(bodys), _p(counts), _p(gt_roots), _p(gt_counts), B, gt_roots.shape[1],
                                            _p(matched), _p(mcounts), _stream()), "smap_register_gt")
    return matched, mcounts


def lift_batch(bodys, counts, det_d, root_d, cams, gt_mode=False, out=None):
    """Batched 3D lifting (test.py:116-134, test_util.py:45-99, post_3d.py).
    det_d [B,14,H,W], root_d [B,H,W] or [B,1,H,W], cams [B,9] float64
    (scale,img_w,img_h,net_w,net_h,f_x,f_y,cx,cy).
    -> pred_2d [B,127,15,4] fp32, pred_3d [B,127,15,4] f64, root_z [B,127] f64 (device).
    gt_mode: the float64 person array of the ground-truth modes (pred_2d comes back as f64)."""
    B = bodys.shape[0]
    if root_d.dim() == 4:
        root_d = root_d[:, 0]
    H, W = det_d.shape[-2:]
    if tuple(det_d.shape) != (B, NL, H, W) or tuple(root_d.shape) != (B, H, W):
        raise ValueError("det_d must be [B,14,H,W] and root_d [B,H,W]")
    dev = bodys.device
    det_d = det_d.to(device=dev, dtype=torch.float32).contiguous()
    root_d = root_d.to(device=dev, dtype=torch.float32).contiguous()
    cams = torch.as_tensor(cams, dtype=torch.float64).to(dev).contiguous()
    if tuple(cams.shape) != (B, 9):
        raise ValueError("cams must be [B,9]")
    if out is not None:            # (p2, p3, rz) to write into: views of one result buffer that goes to the host in ONE copy (pipeline.py)
        p2, p3, rz = out
        want = ((B, MAXP, NJ, 4), torch.float64 if gt_mode else torch.float32), ((B, MAXP, NJ, 4), torch.float64), ((B, MAXP), torch.float64)
        for t, (shp, dt) in zip(out, want):
            if tuple(t.shape) != shp or t.dtype != dt or t.device != dev or not t.is_contiguous():
                raise ValueError("lift_batch(out=...): p2 / p3 / rz tensors of the documented shapes and dtypes on the maps' device")
    else:
        p2 = torch.empty((B, MAXP, NJ, 4), dtype=torch.float64 if gt_mode else torch.float32, device=dev)
        p3 = torch.empty((B, MAXP, NJ, 4), dtype=torch.float64, device=dev)
        rz = torch.empty((B, MAXP), dtype=torch.float64, device=dev)
    fn = _L.load().smap_lift_gt if gt_mode else _L.load().smap_lift
    with torch.cuda.device(dev):
        _L.check(fn(_p(bodys), _p(counts), _p(det_d), _p(root_d), _p(cams), B, H, W, _p(p2), _p(p3), _p(rz), _stream()),
                 "smap_lift_gt" if gt_mode else "smap_lift")
    return p2, p3, rz


def refine_batch(pred_2d, pred_3d, counts, wt, bs, out=None):
    """RefineNet post-refinement (test_util.py:102-131).  wt/bs: 5 folded, transposed
    [in][out] fp32 weight tensors and 5 bias tensors on the device.  out: the [B,127,15,4] f64 tensor to write into."""
    B = pred_2d.shape[0]
    dev = pred_2d.device
    if out is None:
        out = torch.empty((B, MAXP, NJ, 4), dtype=torch.float64, device=dev)
    elif tuple(out.shape) != (B, MAXP, NJ, 4) or out.dtype != torch.float64 or out.device != dev or not out.is_contiguous():
        raise ValueError("refine_batch(out=...): a contiguous [B,127,15,4] float64 tensor on the inputs' device")
    wp = (C.c_void_p * 5)(*[w.data_ptr() for w in wt])
    bp = (C.c_void_p * 5)(*[b.data_ptr() for b in bs])
    gt_mode = pred_2d.dtype == torch.float64       # the f64 pred_2d of lift_batch(gt_mode=True)
    fn = _L.load().smap_refine_gt if gt_mode else _L.load().smap_refine
    with torch.cuda.device(dev):
        _L.check(fn(_p(pred_2d), _p(pred_3d), _p(counts), B, wp, bp, _p(out), _stream()),
                 "smap_refine_gt" if gt_mode else "smap_refine")
    return out


def refine_mlp(x, wt, bs):
    """RefineNet.forward (refinenet.py:34-37): x [N,75] fp32 on the device -> [N,45]."""
    if x.dim() != 2 or x.shape[1] != 75 or x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError("x must be a float32 GPU tensor [N,75]")
    x = x.contiguous()
    y = torch.empty((x.shape[0], 45), dtype=torch.float32, device=x.device)
    wp = (C.c_void_p * 5)(*[w.data_ptr() for w in wt])
    bp = (C.c_void_p * 5)(*[b.data_ptr() for b in bs])
    with torch.cuda.device(x.device):
        _L.check(_L.load().smap_refine_mlp(_p(x), x.shape[0], wp, bp, _p(y), _stream()), "smap_refine_mlp")
    return y


# ---- the reference's two entry points -------------------------------------------------------
def extract(hmsIn):
    """dapalib.extract (association.cpp:34-120): 15 CPU tensors [n_j,3] and 14 CPU tensors [nA,nB]."""
    _check_hms(hmsIn, False)
    peaks, scores = extract_batch(hmsIn.unsqueeze(0))
    peaks, scores = peaks[0].cpu(), scores[0].cpu()
    n = [int(peaks[j, 0, 0]) for j in range(NJ)]
    cands = [peaks[j, 1:1 + n[j]].clone() for j in range(NJ)]
    pafs = [scores[l, :n[JOINT_PAIRS[2 * l]], :n[JOINT_PAIRS[2 * l + 1]]].clone() for l in range(NL)]
    return cands, pafs


def connect(hmsIn, rDepth, rootIdx=2, distFlag=True):
    """dapalib.connect (association.cpp:123-233): CPU fp32 [P,15,4], or an empty [0] tensor."""
    H, W = _check_hms(hmsIn, False)
    if not isinstance(rDepth, torch.Tensor) or tuple(rDepth.shape) != (H, W):
        raise ValueError(f"rDepth must be a [{H},{W}] tensor")
    bodys, counts = connect_batch(hmsIn.unsqueeze(0), rDepth.unsqueeze(0), rootIdx, distFlag)
    P = int(counts.cpu()[0])
    if P == 0:
        return torch.empty((0,))
    return bodys[0, :P].cpu()
