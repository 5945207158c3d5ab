"""Inference-relevant part of the reference's exps/stage3_root2/config.py (Config class, :10-71).
Training-only entries (SOLVER, DATALOADER, tensorboard) are omitted: training is out of scope."""
import os
import os.path as osp

from smap_amd.utils.attrdict import AttrDict as edict
from dataset.data_settings import load_dataset


class Config:
    ROOT_DIR = os.environ.get("PROJECT_HOME", osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
    OUTPUT_DIR = osp.join(ROOT_DIR, "model_logs", osp.split(osp.split(osp.realpath(__file__))[0])[1])
    TEST_DIR = osp.join(OUTPUT_DIR, "log_dir")

    DATASET = edict()
    DATASET.NAME = "MIX"
    dataset = load_dataset(DATASET.NAME)
    DATASET.KEYPOINT = dataset.KEYPOINT
    DATASET.PAF = dataset.PAF
    DATASET.ROOT_IDX = dataset.ROOT_IDX
    DATASET.MAX_PEOPLE = 20

    INPUT = edict()
    INPUT.NORMALIZE = True
    INPUT.MEANS = [0.406, 0.456, 0.485]   # bgr
    INPUT.STDS = [0.225, 0.224, 0.229]
    INPUT_SHAPE = dataset.INPUT_SHAPE
    OUTPUT_SHAPE = dataset.OUTPUT_SHAPE

    MODEL = edict()
    MODEL.STAGE_NUM = 3
    MODEL.UPSAMPLE_CHANNEL_NUM = 256
    MODEL.DEVICE = "cuda"
    MODEL.WEIGHT = None

    LOSS = edict()
    LOSS.OHKM = True
    LOSS.TOPK = 8
    LOSS.COARSE_TO_FINE = True

    RUN_EFFICIENT = False

    TEST = edict()
    TEST.IMG_PER_GPU = 16


config = Config()
cfg = config
