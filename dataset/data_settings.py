"""Skeleton / geometry constants of the MIX setting (reference: dataset/data_settings.py:11-41).
Values are data and must equal the reference's; they are derived here from the skeleton tree
instead of being listed: 15 key points, 14 limbs, mirror tables, 512x832 input, stride 4."""
import os

from smap_amd.utils.attrdict import AttrDict

# joint ids: 0 neck, 1 head, 2 pelvis, then (shoulder, elbow, wrist) and (hip, knee, ankle) for the
# left side (3-5, 6-8) and the right side (9-11, 12-14)
_LEFT_ARM, _LEFT_LEG, _RIGHT_ARM, _RIGHT_LEG = (3, 4, 5), (6, 7, 8), (9, 10, 11), (12, 13, 14)


def _limbs():
    chain = lambda root, js: [[a, b] for a, b in zip((root,) + js[:-1], js)]
    return [[0, 1], [0, 2]] + chain(0, _RIGHT_ARM) + chain(0, _LEFT_ARM) + chain(2, _RIGHT_LEG) + chain(2, _LEFT_LEG)


def _mirror_joints():
    m = list(range(15))
    for l, r in zip(_LEFT_ARM + _LEFT_LEG, _RIGHT_ARM + _RIGHT_LEG):
        m[l], m[r] = r, l
    return m


def _mirror_paf_channels(limbs, mirror):
    """Channel 2l / 2l+1 (x / y of limb l) maps to the channels of the mirrored limb."""
    index = {tuple(l): i for i, l in enumerate(limbs)}
    out = []
    for a, b in limbs:
        j = index[(mirror[a], mirror[b])]
        out += [2 * j, 2 * j + 1]
    return out


def _build():
    d = AttrDict(NAME="MIX", ROOT_IDX=2, STRIDE=4, INPUT_SHAPE=(512, 832))
    limbs, mirror = _limbs(), _mirror_joints()
    d.KEYPOINT = AttrDict(NUM=15, FLIP_ORDER=mirror)
    d.PAF = AttrDict(VECTOR=limbs, NUM=len(limbs), FLIP_CHANNEL=_mirror_paf_channels(limbs, mirror), LINE_WIDTH_THRE=1)
    d.OUTPUT_SHAPE = tuple(s // d.STRIDE for s in d.INPUT_SHAPE)
    d.WIDTH_HEIGHT_RATIO = d.INPUT_SHAPE[1] / d.INPUT_SHAPE[0]
    d.PREFIX = os.environ.get("PROJECT_HOME", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    # where the annotated datasets live (data_settings.py:45-54); only the ground-truth modes of test.py read them
    for name, folder, ann in (("COCO", "coco2017", "coco_keypoints_train2017"), ("MUCO", "MuCo", "MuCo"),
                              ("CMUP", "Panoptic", "Panoptic"), ("H36M", "Human3.6M", "H36M")):
        d[name + "_ROOT_PATH"] = os.path.join(d.PREFIX, "data", folder)
        d[name + "_JSON_PATH"] = os.path.join(d[name + "_ROOT_PATH"], "annotations", ann + ".json")
    d.USED_3D_DATASETS = ["MUCO"]
    return d


MIX = _build()


def load_dataset(name):
    return MIX if "MIX" in name else None
