"""GPU pre-processing for `run_inference`: the reference decodes, resizes,
pads and normalises every image on the host (dataset/custom_dataset.py:27-68); here only the
decode stays on the host, the uint8 image is uploaded as is (3 B/pixel instead of 12) and one HIP
kernel (`smap_preprocess`) writes the letter-boxed, normalised fp32 frame straight into the batch.
`letterbox_geometry` is shared with the host path (dataset/custom_dataset.py) so both agree on the
resized size and the padding offsets."""
import ctypes as C

import warnings

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


def _cv_axis(n_out, scale, n_in):
    """OpenCV's per-axis set-up for 8-bit INTER_LINEAR (modules/imgproc/src/resize.cpp, cv::resize): source index and the
    two 11-bit fixed-point coefficients of every destination index."""
    f32 = np.float32
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(f32)       # (float)((d + 0.5) * scale - 0.5)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(f32)).astype(f32)
    lo, hi = s < 0, s >= n_in - 1
    f = np.where(lo | hi, f32(0), f).astype(f32)
    s = np.where(lo, 0, np.where(hi, n_in - 1, s))
    c0 = np.rint((f32(1) - f) * f32(2048)).astype(np.int32)                           # saturate_cast<short>: cvRound, half to even
    c1 = np.rint(f * f32(2048)).astype(np.int32)
    return s, np.minimum(s + 1, n_in - 1), c0, c1


def resize_linear_u8(img, nh, nw, fx=None, fy=None):
    """cv2.resize(img, (0, 0), fx=fx, fy=fy) -- INTER_LINEAR on 8-bit images -- restated from OpenCV's published algorithm,
    operation by operation (cv2 itself is not installed: parity unpinned by execution):
      * dsize = cvRound(src * f) (the caller's nh, nw); source coordinate (d + 0.5) / f - 0.5 in double, then float;
      * 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS), horizontal pass in int32:  H = S[s] * a0 + S[s+1] * a1;
      * vertical pass  dst = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2
        (VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>);
      * an exact 2x shrink is the 2x2 box mean (cv::resize turns INTER_LINEAR into INTER_AREA there).
    fx / fy default to dsize / ssize (what cv2.resize does when dsize, not fx / fy, is given).  The HIP kernel
    (csrc/assoc.hip::preprocess_kernel) computes the same integers."""
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    sx = 1.0 / (fx if fx is not None else nw / w)
    sy = 1.0 / (fy if fy is not None else nh / h)
    if sx == 2.0 and sy == 2.0:
        y0, x0 = np.minimum(np.arange(nh) * 2, h - 1), np.minimum(np.arange(nw) * 2, w - 1)
        y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
        i = img.astype(np.int32)
        return ((i[y0][:, x0] + i[y0][:, x1] + i[y1][:, x0] + i[y1][:, x1] + 2) >> 2).astype(np.uint8)
    ys0, ys1, b0, b1 = _cv_axis(nh, sy, h)
    xs0, xs1, a0, a1 = _cv_axis(nw, sx, w)
    i = img.astype(np.int32)
    a0, a1 = a0[None, :, None], a1[None, :, None]
    h0 = i[ys0][:, xs0] * a0 + i[ys0][:, xs1] * a1
    h1 = i[ys1][:, xs0] * a0 + i[ys1][:, xs1] * a1
    b0, b1 = b0[:, None, None], b1[:, None, None]
    t = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(t, 0, 255).astype(np.uint8)


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
            if isinstance(im, torch.Tensor):
                t = im.contiguous()
            else:
                with warnings.catch_warnings():          # the decoder's frames are read-only views of its bytes: they are only read here
                    warnings.simplefilter("ignore", UserWarning)
                    t = torch.as_tensor(np.ascontiguousarray(im))
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("images must be uint8 HxWx3 (BGR)")
            h0, w0 = int(t.shape[0]), int(t.shape[1])
            scale, (nh, nw, top, left) = letterbox_geometry(w0, h0, net_w, net_h)
            d = t.to(device, non_blocking=True)
            keep.append(d)
            _L.check(lib.smap_preprocess(C.c_void_p(d.data_ptr()), h0, w0, nh, nw, top, left,
                                         C.c_void_p(out[i].data_ptr()), net_h, net_w, mean, std, scale["scale"], scale["scale"], st),
                     "smap_preprocess")
            for k in scales:
                scales[k].append(scale[k])
    return out, scales
