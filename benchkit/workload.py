"""Synthetic workloads for bench.py and the tests: the model config, association scenes (SURVEY.md 8d config 3:
K in {0,2,8,20} persons), pure-noise maps, and recipe weights whose heads are calibrated to produce detections."""
from types import SimpleNamespace as NS

import numpy as np

NJ, NL = 15, 14
PAIRS = [(0, 1), (0, 2), (0, 9), (9, 10), (10, 11), (0, 3), (3, 4), (4, 5), (2, 12), (12, 13), (13, 14),
         (2, 6), (6, 7), (7, 8)]          # dataset/data_settings.py:28-32

# template skeleton, cm, relative to the neck; joint order of data_settings.py:16-21
TEMPLATE_CM = np.array([
    (0, 0), (0, -26), (0, 48),
    (15, 2), (17, 33), (18, 57), (12, 50), (13, 98), (14, 137),
    (-15, 2), (-17, 33), (-18, 57), (-12, 50), (-13, 98), (-14, 137)], np.float64)


def make_cfg(out_shape=(128, 208)):
    return NS(MODEL=NS(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256),
              DATASET=NS(KEYPOINT=NS(NUM=15), PAF=NS(NUM=14)),
              OUTPUT_SHAPE=tuple(out_shape),
              LOSS=NS(OHKM=True, TOPK=8, COARSE_TO_FINE=True))


def synth_scene(n_person, seed, H=128, W=208, noise=0.02, sigma=1.5, drop=0.1):
    """Gaussian keypoint blobs + unit-vector PAF ribbons + per-person root-depth discs.

    Returns hms [43,H,W] fp32 (already in the /255,/127-scaled range the association
    expects), rdepth [H,W] fp32, and the ground-truth joints [P,15,2] / depths [P]."""
    rng = np.random.default_rng(seed)
    hms = np.zeros((43, H, W), np.float32)
    rdepth = np.full((H, W), 1.0, np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    joints = np.zeros((n_person, NJ, 2))
    depths = np.zeros((n_person,))
    for p in range(n_person):
        d = rng.uniform(0.5, 1.2)                   # root depth Z / f_x / scale: 1 cm = 1/(4d) heat-map px
        ext = 1.0 / (4.0 * d)
        cx = rng.uniform(22 * ext + 6, W - 22 * ext - 6)
        cy = rng.uniform(30 * ext + 6, H - 142 * ext - 6)
        th = rng.normal(0, 0.15)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        j = (TEMPLATE_CM @ R.T) / (d * 4.0) + np.array([cx, cy])
        j += rng.normal(0, 0.3, j.shape)
        joints[p], depths[p] = j, d
        present = rng.random(NJ) > drop
        present[2] = True
        for k in range(NJ):
            if not present[k] or not (3 < j[k, 0] < W - 4 and 3 < j[k, 1] < H - 4):
                present[k] = False
                continue
            g = np.exp(-((xx - j[k, 0]) ** 2 + (yy - j[k, 1]) ** 2) / (2 * sigma ** 2))
            hms[k] = np.maximum(hms[k], g.astype(np.float32))
        for l, (a, b) in enumerate(PAIRS):
            if not (present[a] and present[b]):
                continue
            v = j[b] - j[a]
            n = np.linalg.norm(v)
            if n < 1e-3:
                continue
            u = v / n
            t = (xx - j[a, 0]) * u[0] + (yy - j[a, 1]) * u[1]
            dist = np.abs((xx - j[a, 0]) * u[1] - (yy - j[a, 1]) * u[0])
            m = (t >= -2.0) & (t <= n + 2.0) & (dist <= 2.4)   # samples sit up to ~1.4 px off the true line (+0.5 offset, int round)
            hms[15 + 2 * l][m] = u[0]
            hms[16 + 2 * l][m] = u[1]
        m = (xx - j[2, 0]) ** 2 + (yy - j[2, 1]) ** 2 <= 16
        rdepth[m] = d
    hms += rng.normal(0, noise, hms.shape).astype(np.float32)
    return hms, rdepth, joints, depths


def noise_scene(seed, H=128, W=208, amp=1.0):
    """Pure-noise maps: hundreds of peaks per channel (exercises the 127-peak cap)."""
    rng = np.random.default_rng(seed)
    hms = rng.uniform(0, amp, (43, H, W)).astype(np.float32)
    hms[15:] = rng.uniform(-1, 1, (28, H, W)).astype(np.float32)
    rdepth = rng.uniform(0.2, 1.0, (H, W)).astype(np.float32)
    return hms, rdepth


# ---- recipe weights whose heads produce detections -------------------------------------------------
# Random weights give key-point maps whose channels sit entirely above or below the 0.2*255 NMS threshold
# ("pelvis channel empty => connect returns shape [0]").  For end-to-end runs through the REAL
# backbone the last BatchNorm of the stage-2 key-point head is re-centred per channel so that every key-point
# channel carries ~24 peaks on frame 0 of the seed-1234 batch, and the root-depth head is rescaled so that
# root_d * scale * f_x lands at a few metres.  The numbers come from benchkit/calibrate_heads.py (CPU, fp32 oracle
# forward) and are data, not weights: benchkit/head_calibration.json.
HEAD_KINDS = {"noise": 1.0,     # full-resolution head at full gain: peaks are isolated noise maxima (fragile: the hard case)
              "smooth": 0.1}    # full-resolution head x0.1: maps dominated by the bilinearly upsampled coarse heads
RD_GAIN = 0.0224                # root_d ~ 0.46 -> Z ~ 300 cm with scale * f_x = 650
PEOPLE_CAM = (0.43333333333333335, 1920.0, 1080.0, 832.0, 512.0, 1500.0, 1500.0, 960.0, 540.0)


def people_state_dict(sd, kind="smooth", calibration=None):
    """recipe_state_dict(sd) with the stage-2 heads calibrated as described above (sd: a SMAP state_dict)."""
    import json
    import os
    from .recipe import recipe_state_dict
    out = recipe_state_dict(sd)
    g4 = HEAD_KINDS[kind]
    p = "stage2.upsample.up4.res_conv2.bn."
    out[p + "weight"] = out[p + "weight"] * g4
    out[p + "bias"] = out[p + "bias"] * g4
    q = "stage2.upsample.up4.res_rd_conv2.bn."
    out[q + "weight"] = out[q + "weight"] * RD_GAIN
    out[q + "bias"] = out[q + "bias"] * RD_GAIN
    if calibration is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "head_calibration.json")
        calibration = json.load(open(path))[kind]["kpt_bias"]
    if calibration:
        b = out[p + "bias"].clone()
        b[:NJ] += b.new_tensor(calibration)
        out[p + "bias"] = b
    return out
