"""ctypes binding of oracle/libsmap_oracle.so (numpy in / numpy out).

TEST INFRASTRUCTURE ONLY -- see the header of smap_oracle.c.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package (smap_amd, dapalib, model, exps) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsmap_oracle.so")
NJ, NL, MAXP = 15, 14, 127
REFINE_DIMS = (75, 160, 256, 256, 128, 45)


def build(force=False):
    src = os.path.join(_HERE, "smap_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libsmap_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_int
        _lib.smap_oracle_nms.argtypes = [fp, ip, ip, C.c_float, fp]
        _lib.smap_oracle_paf_score.argtypes = [fp, fp, ip, ip, fp]
        _lib.smap_oracle_group.argtypes = [fp, fp, fp, ip, ip, ip, fp]
        _lib.smap_oracle_group.restype = ip
        _lib.smap_oracle_lift.argtypes = [fp, ip, fp, fp, ip, ip, dp, fp, dp, dp]
        _lib.smap_oracle_refine.argtypes = [fp, dp, ip, C.POINTER(fp), C.POINTER(fp), dp]
        _lib.smap_oracle_sort_depth.argtypes = [fp, ip, C.POINTER(C.c_int), fp]
        _lib.smap_oracle_connect.argtypes = [fp, fp, ip, ip, ip, ip, fp, fp, fp]
        _lib.smap_oracle_connect.restype = ip
        _lib.smap_oracle_register_gt.argtypes = [fp, ip, fp, ip, fp]
        _lib.smap_oracle_register_gt.restype = ip
        _lib.smap_oracle_lift_gt.argtypes = [fp, ip, fp, fp, ip, ip, dp, dp, dp, dp]
        _lib.smap_oracle_refine_gt.argtypes = [dp, dp, ip, C.POINTER(fp), C.POINTER(fp), dp]
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def nms(hms, thr=0.2):
    hms = _c32(hms)
    _, H, W = hms.shape
    peaks = np.zeros((NJ, MAXP + 1, 3), np.float32)
    lib().smap_oracle_nms(_f(hms), H, W, thr, _f(peaks))
    return peaks


def paf_score(hms, peaks):
    hms, peaks = _c32(hms), _c32(peaks)
    _, H, W = hms.shape
    scores = np.empty((NL, MAXP, MAXP), np.float32)
    lib().smap_oracle_paf_score(_f(hms), _f(peaks), H, W, _f(scores))
    return scores


def group(peaks, scores, rdepth, root_idx=2, dist_flag=True):
    peaks, scores, rdepth = _c32(peaks), _c32(scores), _c32(rdepth)
    bodys = np.zeros((MAXP, NJ, 4), np.float32)
    P = lib().smap_oracle_group(_f(peaks), _f(scores), _f(rdepth), rdepth.shape[1], root_idx,
                                int(dist_flag), _f(bodys))
    return bodys[:P].copy()


def connect(hms, rdepth, root_idx=2, dist_flag=True):
    """dapalib.connect restated: returns (bodys[P,15,4], peaks, scores)."""
    hms, rdepth = _c32(hms), _c32(rdepth)
    _, H, W = hms.shape
    peaks = np.zeros((NJ, MAXP + 1, 3), np.float32)
    scores = np.empty((NL, MAXP, MAXP), np.float32)
    bodys = np.zeros((MAXP, NJ, 4), np.float32)
    P = lib().smap_oracle_connect(_f(hms), _f(rdepth), H, W, root_idx, int(dist_flag),
                                  _f(peaks), _f(scores), _f(bodys))
    return bodys[:P].copy(), peaks, scores


def lift(bodys, det_d, root_d, cam):
    bodys, det_d, root_d = _c32(bodys), _c32(det_d), _c32(root_d)
    cam = np.ascontiguousarray(cam, np.float64)
    P = bodys.shape[0]
    _, H, W = det_d.shape
    p2 = np.zeros((P, NJ, 4), np.float32)
    p3 = np.zeros((P, NJ, 4), np.float64)
    rz = np.zeros((P,), np.float64)
    if P:
        lib().smap_oracle_lift(_f(bodys), P, _f(det_d), _f(root_d), H, W, _d(cam), _f(p2), _d(p3), _d(rz))
    return p2, p3, rz


def refine(pred_2d, pred_3d, weights, biases):
    """weights/biases: 5 BN-folded fp32 arrays ([out,in] / [out])."""
    pred_2d = _c32(pred_2d)
    pred_3d = np.ascontiguousarray(pred_3d, np.float64)
    P = pred_2d.shape[0]
    ws = [_c32(w) for w in weights]
    bs = [_c32(b) for b in biases]
    fp = C.POINTER(C.c_float)
    wp = (fp * 5)(*[_f(w) for w in ws])
    bp = (fp * 5)(*[_f(b) for b in bs])
    out = np.zeros((P, NJ, 4), np.float64)
    if P:
        lib().smap_oracle_refine(_f(pred_2d), _d(pred_3d), P, wp, bp, _d(out))
    return out


def register_gt(bodys, gt_root):
    """register_pred with ground truth (test_util.py:18-42): [G,15,4] fp32 in heat-map pixels."""
    bodys, gt_root = _c32(bodys), _c32(gt_root)
    P, Gn = bodys.shape[0], gt_root.shape[0]
    out = np.zeros((Gn, NJ, 4), np.float32)
    if Gn:
        lib().smap_oracle_register_gt(_f(bodys), P, _f(gt_root), Gn, _f(out))
    return out


def lift_gt(bodys, det_d, root_d, cam):
    """lift() for the ground-truth modes (f64 person array): pred_2d comes back as float64."""
    bodys, det_d, root_d = _c32(bodys), _c32(det_d), _c32(root_d)
    cam = np.ascontiguousarray(cam, np.float64)
    P = bodys.shape[0]
    _, H, W = det_d.shape
    p2 = np.zeros((P, NJ, 4), np.float64)
    p3 = np.zeros((P, NJ, 4), np.float64)
    rz = np.zeros((P,), np.float64)
    if P:
        lib().smap_oracle_lift_gt(_f(bodys), P, _f(det_d), _f(root_d), H, W, _d(cam), _d(p2), _d(p3), _d(rz))
    return p2, p3, rz


def refine_gt(pred_2d, pred_3d, weights, biases):
    pred_2d = np.ascontiguousarray(pred_2d, np.float64)
    pred_3d = np.ascontiguousarray(pred_3d, np.float64)
    P = pred_2d.shape[0]
    ws = [_c32(w) for w in weights]
    bs = [_c32(b) for b in biases]
    fp = C.POINTER(C.c_float)
    wp = (fp * 5)(*[_f(w) for w in ws])
    bp = (fp * 5)(*[_f(b) for b in bs])
    out = np.zeros((P, NJ, 4), np.float64)
    if P:
        lib().smap_oracle_refine_gt(_d(pred_2d), _d(pred_3d), P, wp, bp, _d(out))
    return out


def sort_depth(d):
    """Person order for root depths d (n <= 127): (indices, sorted values), torch.sort(0, False) order."""
    d = _c32(d)
    n = d.shape[0]
    idx = np.zeros((max(n, 1),), np.int32)
    out = np.zeros((max(n, 1),), np.float32)
    lib().smap_oracle_sort_depth(_f(d), n, idx.ctypes.data_as(C.POINTER(C.c_int)), _f(out))
    return idx[:n], out[:n]
