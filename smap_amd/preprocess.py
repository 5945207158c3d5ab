"""GPU pre-processing for `run_inference`: the reference decodes, resizes,
pads and normalises every image on the host (dataset/custom_dataset.py:27-68); here only the
decode stays on the host, the uint8 image is uploaded as is (3 B/pixel instead of 12) and one HIP
kernel (`smap_preprocess`) writes the letter-boxed, normalised fp32 frame straight into the batch.
`letterbox_geometry` is shared with the host path (dataset/custom_dataset.py) so both agree on the
resized size and the padding offsets."""
import ctypes as C

import numpy as np
import torch

from . import lib as _L


def letterbox_geometry(w0, h0, net_w=832, net_h=512):
    """scale dict + (nh, nw, top, left) of the resized image inside the canvas
    (custom_dataset.py:44-66: scale = min ratio, pad the short side symmetrically)."""
    s = min(net_w / w0, net_h / h0)
    nw, nh = int(round(w0 * s)), int(round(h0 * s))
    if nw < net_w:
        left = (net_w - nw) // 2
        top = (net_h - nh) // 2 if nh < net_h else 0
    else:
        left, top = 0, (net_h - nh) // 2
    scale = dict(scale=s, img_width=w0, img_height=h0, net_width=net_w, net_height=net_h)
    return scale, (nh, nw, top, left)


def resize_bilinear_u8(img, nh, nw):
    """Host statement of the resize the HIP kernel performs (one rounding per written fp32 operation,
    no FMA contraction): half-pixel-centre bilinear, the sampling rule of cv2.INTER_LINEAR /
    F.interpolate(align_corners=False, antialias=False); result rounded half-to-even to uint8."""
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    f32 = np.float32

    def axis(n_in, n_out):
        scale = f32(n_in) / f32(n_out)
        src = scale * (np.arange(n_out, dtype=f32) + f32(0.5)) - f32(0.5)
        src = np.maximum(src, f32(0))
        i0 = src.astype(np.int64)
        i1 = i0 + (i0 < n_in - 1)
        l1 = (src - i0.astype(f32)).astype(f32)
        return i0, i1, (f32(1) - l1).astype(f32), l1

    y0, y1, ly0, ly1 = axis(h, nh)
    x0, x1, lx0, lx1 = axis(w, nw)
    a = img[y0][:, x0].astype(f32)
    b = img[y0][:, x1].astype(f32)
    c = img[y1][:, x0].astype(f32)
    d = img[y1][:, x1].astype(f32)
    lx0, lx1 = lx0[None, :, None], lx1[None, :, None]
    ly0, ly1 = ly0[:, None, None], ly1[:, None, None]
    t = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * c + lx1 * d)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def preprocess_batch(images, means, stds, device, net_w=832, net_h=512):
    """images: list of uint8 HxWx3 BGR arrays/tensors.  Returns (imgs [B,3,net_h,net_w] fp32 on `device`,
    scales: dict of lists as the DataLoader would collate them)."""
    lib = _L.load()
    B = len(images)
    out = torch.empty((B, 3, net_h, net_w), dtype=torch.float32, device=device)
    mean = (C.c_float * 3)(*means)
    std = (C.c_float * 3)(*stds)
    scales = {k: [] for k in ("scale", "img_width", "img_height", "net_width", "net_height")}
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    keep = []
    with torch.cuda.device(device):
        for i, im in enumerate(images):
            t = torch.as_tensor(np.ascontiguousarray(im) if not isinstance(im, torch.Tensor) else im.contiguous())
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("images must be uint8 HxWx3 (BGR)")
            h0, w0 = int(t.shape[0]), int(t.shape[1])
            scale, (nh, nw, top, left) = letterbox_geometry(w0, h0, net_w, net_h)
            d = t.to(device, non_blocking=True)
            keep.append(d)
            _L.check(lib.smap_preprocess(C.c_void_p(d.data_ptr()), h0, w0, nh, nw, top, left,
                                         C.c_void_p(out[i].data_ptr()), net_h, net_w, mean, std, st), "smap_preprocess")
            for k in scales:
                scales[k].append(scale[k])
    return out, scales
