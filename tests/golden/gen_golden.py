#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the Python reference.

Runs only in the authoring container (needs /root/reference); the GPU box and
the test-suite only ever read the .npz files this writes next to itself.
Nothing of the reference's source travels: fixtures are inputs + outputs.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

What is executed from the reference (imported, unmodified):
  * model.smap.SMAP              (model/smap.py:313-421)        -> backbone_small.npz
  * model.refinenet.RefineNet    (model/refinenet.py:29-37)     -> refine.npz
  * exps/stage3_root2/test_util.py: register_pred, generate_relZ, gen_3d_pose,
    lift_and_refine_3d_pose (:18-131) and lib/utils/post_3d.py  -> lift.npz, refine.npz

Environment accommodations (none of them changes reference arithmetic):
  * `easydict` is not installed: the reference's config.py only uses it as an
    attribute-access dict, so a 6-line attribute dict is registered under that
    module name before import.
  * numpy>=2 removed the aliases np.float / np.int that test_util.py uses
    (:25-26,36,62-63; post_3d.py:12,20); they are re-bound to float / int.
  * env PROJECT_HOME must exist for config.py:13.
  * cv2 is absent: the two cv2.resize(INTER_NEAREST) calls of test.py:123-126
    are outside the imported functions; the x4 nearest upsample is produced
    here with np.repeat (exact for an integer ratio).  That one step is the
    only restated glue in these fixtures and is recorded as such in DESIGN.md.
"""
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SMAP_REFERENCE", "/root/reference")


def _install_reference():
    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    m = types.ModuleType("easydict")
    m.EasyDict = EasyDict
    sys.modules["easydict"] = m
    np.float = float
    np.int = int
    os.environ.setdefault("PROJECT_HOME", "/tmp/smap_project_home")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "exps", "stage3_root2"))
    sys.path.insert(0, HERE)


sys.path.insert(0, HERE)
from recipe import key_seed, recipe_state_dict  # noqa: E402  (tests/golden/recipe.py)


def make_cfg(out_shape):
    from easydict import EasyDict as edict
    cfg = edict()
    cfg.MODEL = edict(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256)
    cfg.DATASET = edict(KEYPOINT=edict(NUM=15), PAF=edict(NUM=14))
    cfg.OUTPUT_SHAPE = out_shape
    cfg.LOSS = edict(OHKM=True, TOPK=8, COARSE_TO_FINE=True)
    return cfg


def gen_backbone():
    from model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    sd = recipe_state_dict(net.state_dict())
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 3, 64, 96, generator=g)
    with torch.no_grad():
        hms, det_d, root_d = net(x)
    keys = sorted(sd.keys())
    np.savez_compressed(
        os.path.join(HERE, "backbone_small.npz"),
        x=x.numpy(), hms=hms.numpy(), det_d=det_d.numpy(), root_d=root_d.numpy(),
        n_keys=np.int64(len(keys)),
        key_digest=np.int64(zlib.crc32("\n".join(keys).encode())),
        n_params=np.int64(sum(p.numel() for p in net.parameters())),
    )
    # default-init statistics: pins the module-construction RNG order
    torch.manual_seed(0)
    net0 = SMAP(make_cfg((128, 208)))
    sd0 = net0.state_dict()
    probe = ["top.conv.conv.weight", "stage0.downsample.layer1.0.conv_bn_relu1.conv.weight",
             "stage1.upsample.up2.skip2.conv.weight", "stage2.upsample.up4.res_rd_conv2.conv.weight",
             "stage2.upsample.up4.res_conv2.conv.bias"]
    np.savez_compressed(
        os.path.join(HERE, "default_init.npz"),
        keys=np.array(probe),
        sums=np.array([float(sd0[k].double().sum()) for k in probe]),
        abssums=np.array([float(sd0[k].double().abs().sum()) for k in probe]),
        all_keys=np.array(list(sd0.keys())),
        all_shapes=np.array([",".join(map(str, sd0[k].shape)) for k in sd0.keys()]),
    )
    print("backbone_small: hms", tuple(hms.shape), "absmax", float(hms.abs().max()),
          "keys", len(keys))


def synth_people(rng, P, H=128, W=208):
    """Synthetic dapalib.connect-style output [P,15,4] in heat-map pixels."""
    bodys = np.zeros((P, 15, 4), np.float32)
    for i in range(P):
        cx, cy = rng.uniform(20, W - 20), rng.uniform(25, H - 25)
        for j in range(15):
            if j != 2 and rng.random() < 0.2:
                continue  # missing joint
            bodys[i, j, 0] = np.clip(cx + rng.normal(0, 9), 0.5, W - 0.5)
            bodys[i, j, 1] = np.clip(cy + rng.normal(0, 12), 0.5, H - 0.5)
            bodys[i, j, 3] = rng.uniform(0.25, 1.0)
    return bodys


def ref_lift(bodys_hm, det_d, root_d, scale, refine_model=None):
    """test.py:116-137 with the reference's own functions."""
    import test_util as tu
    pred = torch.from_numpy(bodys_hm.copy())
    if len(pred) > 0:
        pred[:, :, :2] *= 4                                   # test.py:117
        pred = pred.numpy()
    pafs_3d = det_d.transpose(1, 2, 0)
    paf_up = np.repeat(np.repeat(pafs_3d, 4, axis=0), 4, axis=1)   # cv2 INTER_NEAREST x4
    root_up = np.repeat(np.repeat(root_d, 4, axis=0), 4, axis=1)
    pred = tu.register_pred(pred, None)
    if len(pred) == 0:
        return None
    rdepth = tu.generate_relZ(pred, paf_up, root_up, scale)
    p3d = tu.gen_3d_pose(pred, rdepth, scale)
    out = dict(pred_2d=np.asarray(pred), pred_3d=p3d, root_d=rdepth)
    if refine_model is not None:
        out["refined"] = tu.lift_and_refine_3d_pose(pred, p3d, refine_model, device="cpu")
    return out


def gen_lift_and_refine():
    from model.refinenet import RefineNet
    from fixture_maps import expand
    rnet = RefineNet().eval()
    rsd = recipe_state_dict(rnet.state_dict())
    rnet.load_state_dict(rsd)
    rng = np.random.default_rng(20260926)
    cases = {}
    n = 0
    for P, (img_w, img_h) in [(1, (1920, 1080)), (3, (640, 480)), (7, (2048, 2048)),
                              (20, (1280, 720)), (5, (832, 512))]:
        bodys = synth_people(rng, P)
        det_c = rng.normal(0, 8, (14, 16, 26)).astype(np.float32)
        root_c = rng.uniform(0.5, 3.0, (1, 16, 26)).astype(np.float32)
        det_d = expand(det_c, 0.05)
        root_d = expand(root_c, 0.002)[0]
        s = min(832 / img_w, 512 / img_h)
        scale = {"scale": np.asarray(s), "img_width": np.asarray(img_w), "img_height": np.asarray(img_h),
                 "net_width": np.asarray(832), "net_height": np.asarray(512)}
        scale["f_x"] = scale["img_width"]; scale["f_y"] = scale["img_width"]      # test.py:100-103
        scale["cx"] = scale["img_width"] / 2; scale["cy"] = scale["img_height"] / 2
        with torch.no_grad():
            r = ref_lift(bodys, det_d, root_d, scale, rnet)
        cam = np.array([s, img_w, img_h, 832, 512, float(scale["f_x"]), float(scale["f_y"]),
                        float(scale["cx"]), float(scale["cy"])], np.float64)
        pre = f"c{n}_"
        cases.update({pre + "bodys": bodys, pre + "det_c": det_c, pre + "root_c": root_c,
                      pre + "cam": cam, pre + "pred_2d": r["pred_2d"], pre + "pred_3d": r["pred_3d"],
                      pre + "root_z": r["root_d"], pre + "refined": r["refined"]})
        n += 1
    cases["n_cases"] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, "lift.npz"), **cases)
    # RefineNet forward by itself (eval-mode BN, refinenet.py:29-37)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(33, 75, generator=g) * 20
    with torch.no_grad():
        y = rnet(x)
    np.savez_compressed(os.path.join(HERE, "refine.npz"), x=x.numpy(), y=y.numpy(),
                        n_keys=np.int64(len(rsd)))
    print("lift: cases", n, " refine: y absmax", float(y.abs().max()))


if __name__ == "__main__":
    _install_reference()
    gen_lift_and_refine()
    gen_backbone()
