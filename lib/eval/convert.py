"""MuPoTS evaluation hand-off (reference: lib/eval/convert.py:5-85).

`convert(path)` turns the result JSON of `test.py -t generate_result` into the two MATLAB files the
reference's evaluation scripts read (`mupots_smap.m`): `pose3d.mat` {'preds_3d_kpt': {image: [P,15,4]}}
in millimetres and `pose2d.mat` {'preds_2d_kpt': {image: [P,15,4]}} in original-image pixels.

Per frame (convert.py:15-79):
  * the sequence id `TS<k>` in the image path fixes the original resolution (k < 6: 2048x2048,
    6..20: 1920x1080, otherwise NotImplementedError);
  * 2D joints go back from the 832x512 network frame to the image: subtract the letter-box pad
    (`(crop - size*scale) // 2`, horizontal pad wins when both exist), divide by the scale;
  * every 3D joint is re-projected from its 2D position and predicted depth through the GROUND-TRUTH
    intrinsics of the frame, `Z * K^-1 [u v 1]^T`, with K = [[f,0,cx],[0,f,cy],[0,0,1]] and
    f = gt[0,0,4] (the reference uses the same focal length on both axes), unless the 2D score is 0;
  * centimetres -> millimetres for the three coordinates (the score column is left alone).

The record keys follow the reference's reader ('pred', 'gt'); the writer of this repository and of
the reference (`save_result`, test_util.py:146-158) calls them 'pred_3d' / 'gt_3d', so both are accepted."""
import json
import os

import numpy as np

CROP_X, CROP_Y = 832, 512


def sequence_resolution(image_path):
    """(relative name from 'TS' on, width, height) of a MuPoTS frame (convert.py:16-24)."""
    name = image_path[image_path.index("TS"):]
    ts = int(name[2:name.index("/")])
    if ts < 6:
        return name, 2048, 2048
    if ts <= 20:
        return name, 1920, 1080
    raise NotImplementedError("unknown MuPoTS sequence TS%d" % ts)


def unletterbox(pred_2d, width, height):
    """Network-frame pixels -> original-image pixels (convert.py:42-59); returns a new float64 array."""
    scale = min(CROP_X / float(width), CROP_Y / float(height))
    adj = np.zeros(2)
    if height * scale < CROP_Y:
        adj = np.array([0.0, (CROP_Y - height * scale) // 2])
    if width * scale < CROP_X:
        adj = np.array([(CROP_X - width * scale) // 2, 0.0])
    out = np.array(pred_2d, dtype=np.float64)
    if out.size:
        out[:, :, :2] = (out[:, :, :2] - adj) / scale
    return out


def reproject(pred_3d, pred_2d_img, K):
    """Z * K^-1 [u v 1]^T per joint; joints whose 2D score is 0 keep their prediction (convert.py:61-75)."""
    out = np.array(pred_3d, dtype=np.float64)
    iK = np.linalg.inv(K)
    n = min(len(out), len(pred_2d_img))
    for ih in range(n):
        z = out[ih, :, 2].copy()
        uv1 = np.concatenate([pred_2d_img[ih, :, :2], np.ones((out.shape[1], 1))], axis=1)     # [15,3]
        xyz = np.einsum("j,rk,jk->jr", z, iK, uv1)
        keep = pred_2d_img[ih, :, 3] != 0
        out[ih, keep, :3] = xyz[keep]
    return out


def convert_records(pairs_3d):
    """{image: [P,15,4] mm}, {image: [P,15,4] px}, {image: gt mm} for a list of '3d_pairs' records."""
    pose3d, pose2d, gt3d = {}, {}, {}
    for rec in pairs_3d:
        name, width, height = sequence_resolution(rec["image_path"])
        pred_3d = np.array(rec["pred"] if "pred" in rec else rec["pred_3d"], dtype=np.float64)
        gt_3d = np.array(rec["gt"] if "gt" in rec else rec["gt_3d"], dtype=np.float64)
        f, cx, cy = gt_3d[0, 0, 4], gt_3d[0, 0, 5], gt_3d[0, 0, 6]
        K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], dtype=np.float64)
        p2 = unletterbox(rec["pred_2d"], width, height)
        p3 = reproject(pred_3d, p2, K) * 10.0
        p3[:, :, 3] /= 10.0
        pose3d[name], pose2d[name], gt3d[name] = p3, p2, gt_3d * 10.0
    return pose3d, pose2d, gt3d


def convert(path="", out_dir="."):
    """Reads the result JSON at `path`, writes `pose3d.mat` and `pose2d.mat` into `out_dir` (the
    reference writes into the working directory, convert.py:81-83) and returns the two dicts."""
    import scipy.io as scio
    with open(path, "r") as f:
        data = json.load(f)
    pose3d, pose2d, _ = convert_records(data["3d_pairs"])
    scio.savemat(os.path.join(out_dir, "pose3d.mat"), {"preds_3d_kpt": pose3d})
    scio.savemat(os.path.join(out_dir, "pose2d.mat"), {"preds_2d_kpt": pose2d})
    return pose3d, pose2d


if __name__ == "__main__":
    import sys
    convert(sys.argv[1] if len(sys.argv) > 1 else "xxx.json")
