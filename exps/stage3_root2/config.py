"""Run configuration of the stage3_root2 experiment, inference subset (reference:
exps/stage3_root2/config.py:10-71; SOLVER / DATALOADER / tensorboard entries belong to training,
which is out of scope).  `cfg` has the attribute layout `test.py`, `SMAP(cfg)` and the dataset read."""
import os
import os.path as osp

from smap_amd.utils.attrdict import AttrDict
from dataset.data_settings import load_dataset

_HERE = osp.dirname(osp.realpath(__file__))


def _make():
    ds = load_dataset("MIX")
    root = os.environ.get("PROJECT_HOME", osp.dirname(osp.dirname(_HERE)))
    out_dir = osp.join(root, "model_logs", osp.basename(_HERE))
    return AttrDict(
        ROOT_DIR=root, OUTPUT_DIR=out_dir, TEST_DIR=osp.join(out_dir, "log_dir"),
        DATASET=AttrDict(NAME="MIX", KEYPOINT=ds.KEYPOINT, PAF=ds.PAF, ROOT_IDX=ds.ROOT_IDX, MAX_PEOPLE=20),
        dataset=ds,
        INPUT=AttrDict(NORMALIZE=True, MEANS=[0.406, 0.456, 0.485], STDS=[0.225, 0.224, 0.229]),   # BGR order
        INPUT_SHAPE=ds.INPUT_SHAPE, OUTPUT_SHAPE=ds.OUTPUT_SHAPE,
        MODEL=AttrDict(STAGE_NUM=3, UPSAMPLE_CHANNEL_NUM=256, DEVICE="cuda", WEIGHT=None),
        LOSS=AttrDict(OHKM=True, TOPK=8, COARSE_TO_FINE=True),     # read by SMAP.__init__ only
        RUN_EFFICIENT=False,
        DATALOADER=AttrDict(NUM_WORKERS=int(os.environ.get("SMAP_NUM_WORKERS", 0))),
        # generate_result / generate_train read the MuPoTS annotations from here (config.py:67-70)
        TEST=AttrDict(IMG_PER_GPU=16, ROOT_PATH=os.environ.get("SMAP_TEST_ROOT", "/data/MultiPersonTestSet"),
                      JSON_PATH=osp.join(os.environ.get("SMAP_TEST_ROOT", "/data/MultiPersonTestSet"), "M3E_gt.json")),
    )


cfg = config = _make()
