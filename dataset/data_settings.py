"""Constants of the reference's dataset/data_settings.py (MIX): 15 key points, 14 limbs,
flip tables, 512x832 network input, stride 4."""
import os

from smap_amd.utils.attrdict import AttrDict as edict


class MIX:
    NAME = "MIX"
    KEYPOINT = edict()
    KEYPOINT.NUM = 15
    # 0 neck 1 head 2 pelvis 3-5 left shoulder/elbow/wrist 6-8 left hip/knee/ankle
    # 9-11 right shoulder/elbow/wrist 12-14 right hip/knee/ankle  (data_settings.py:16-21)
    KEYPOINT.FLIP_ORDER = [0, 1, 2, 9, 10, 11, 12, 13, 14, 3, 4, 5, 6, 7, 8]
    ROOT_IDX = 2
    PAF = edict()
    PAF.VECTOR = [[0, 1], [0, 2], [0, 9], [9, 10], [10, 11], [0, 3], [3, 4], [4, 5],
                  [2, 12], [12, 13], [13, 14], [2, 6], [6, 7], [7, 8]]
    PAF.FLIP_CHANNEL = [0, 1, 2, 3, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7, 8, 9,
                        22, 23, 24, 25, 26, 27, 16, 17, 18, 19, 20, 21]
    PAF.NUM = len(PAF.VECTOR)
    PAF.LINE_WIDTH_THRE = 1
    INPUT_SHAPE = (512, 832)
    STRIDE = 4
    OUTPUT_SHAPE = (INPUT_SHAPE[0] // STRIDE, INPUT_SHAPE[1] // STRIDE)
    WIDTH_HEIGHT_RATIO = INPUT_SHAPE[1] / INPUT_SHAPE[0]
    PREFIX = os.environ.get("PROJECT_HOME", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_dataset(name):
    return MIX if "MIX" in name else None
