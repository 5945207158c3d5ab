"""Result assembly for the inference entry point (reference: exps/stage3_root2/test_util.py).

The reference's per-frame numpy post-process -- register_pred (:18-42), generate_relZ (:60-86),
chain_bones (:45-57), gen_3d_pose (:89-99), lift_and_refine_3d_pose (:102-131) -- runs here as
batched HIP kernels (`dapalib.lift_batch`, `dapalib.refine_batch`, include/smap_hip.h) on the
whole batch at once; this module keeps the host-side pieces: camera defaults, the flip-TTA
merge and the `3d_pairs` record schema of save_result (:146-158)."""
import numpy as np
import torch

import dapalib
from smap_amd.records import frame_record, train_records


def default_cams(scales, n):
    """[n,9] float64: scale,img_w,img_h,net_w,net_h,f_x,f_y,cx,cy with the reference's defaults for
    unknown cameras (test.py:97-103): f_x = f_y = img_width, principal point at the image centre."""
    g = lambda k: np.asarray(scales[k], dtype=np.float64).reshape(-1)
    cams = np.zeros((n, 9), np.float64)
    cams[:, 0], cams[:, 1], cams[:, 2] = g("scale"), g("img_width"), g("img_height")
    cams[:, 3], cams[:, 4] = g("net_width"), g("net_height")
    cams[:, 5] = cams[:, 6] = cams[:, 1]
    cams[:, 7], cams[:, 8] = cams[:, 1] / 2, cams[:, 2] / 2
    return cams


def merge_flip(outputs_2d, outputs_2d_flip, cfg):
    """Flip-TTA merge of test.py:55-70 (in place on outputs_2d): channel-permuted sum, PAF-x
    negated, and only the PAF channels averaged (key-point channels stay a SUM of the passes)."""
    kpt = cfg.DATASET.KEYPOINT.NUM
    pair = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [x + kpt for x in cfg.DATASET.PAF.FLIP_CHANNEL]
    sign = torch.ones(len(pair), device=outputs_2d.device)
    sign[kpt::2] = -1.0
    flipped = torch.flip(outputs_2d_flip, dims=[-1])[:, pair]
    outputs_2d += flipped * sign.view(1, -1, 1, 1)
    outputs_2d[:, kpt:] *= 0.5
    return outputs_2d


def poses_from_outputs(outputs_2d, outputs_3d, outputs_rd, cams, cfg, refine_weights=None):
    """test.py:106-140 for the whole batch on the device.  Returns host arrays
    (pred_2d [B,127,15,4] f32, pred_3d [B,127,15,4] f64, root_d [B,127] f64, counts [B])."""
    dapalib.scale_hms_(outputs_2d)                                                    # test.py:111-112
    bodys, counts = dapalib.connect_batch(outputs_2d, outputs_rd, cfg.DATASET.ROOT_IDX, distFlag=True)
    p2, p3, rz = dapalib.lift_batch(bodys, counts, outputs_3d, outputs_rd, cams)
    if refine_weights is not None:
        p3 = dapalib.refine_batch(p2, p3, counts, *refine_weights)
    return p2.cpu().numpy(), p3.cpu().numpy(), rz.cpu().numpy(), counts.cpu().numpy()


def save_result(pred_bodys_2d, pred_bodys_3d, gt_bodys, pred_rdepths, img_path, result):
    """One per-frame record of result['3d_pairs'] (test_util.py:146-158)."""
    result["3d_pairs"].append(frame_record(pred_bodys_2d, pred_bodys_3d, pred_rdepths, img_path, gt_bodys))


def save_result_for_train_refine(pred_bodys_2d, pred_bodys_3d, gt_bodys, pred_rdepths, result, root_n=2):
    """One record per matched person, the RefineNet training pairs (test_util.py:134-143)."""
    result["3d_pairs"].extend(train_records(pred_bodys_2d, pred_bodys_3d, pred_rdepths, gt_bodys, root_n))
