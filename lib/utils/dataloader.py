"""Test-time data loading for the ground-truth modes (reference: lib/utils/dataloader.py:68-113).

`get_test_loader(cfg, num_gpu, local_rank, stage)`: the annotated dataset split in contiguous blocks of
ceil(N / num_gpu) frames per rank (dataloader.py:80-85), read sequentially in batches of cfg.TEST.IMG_PER_GPU;
a batch is (images [B,3,512,832], annotations [B,MAX_PEOPLE,15,C], tuple of image paths, tuple of scale dicts)
(BatchCollator, dataloader.py:98-106).  The training loader (get_train_loader) is outside this repository."""
import math

import torch
from torch.utils.data import DataLoader, Subset

from dataset.base_dataset import JointDataset


def collate_test(batch):
    images, meta, paths, scales = zip(*batch)
    return torch.stack(images, 0), torch.stack(meta, 0), tuple(paths), tuple(scales)


def get_test_loader(cfg, num_gpu, local_rank, stage, use_augmentation=False, with_mds=False):
    if cfg.DATASET.NAME != "MIX":
        raise NameError("Dataset is not defined!", cfg.DATASET.NAME)
    dataset = JointDataset(cfg, stage, None, use_augmentation, with_mds)
    n = len(dataset)
    per = math.ceil(n / num_gpu)
    st = local_rank * per
    ed = min(n, st + per)
    workers = int(cfg.get("DATALOADER", {}).get("NUM_WORKERS", 0)) if hasattr(cfg, "get") else 0
    return DataLoader(Subset(dataset, range(st, ed)), batch_size=cfg.TEST.IMG_PER_GPU, shuffle=False, drop_last=False,
                      num_workers=workers, collate_fn=collate_test)
