"""Reader of the RefineNet training pairs that `test.py -t generate_train` writes (reference:
dataset/p2p_dataset.py:9-40) -- the consumer side of that file format.

Item i: (inp [75] fp32, gt [45] fp32).  `inp` is 15 x (2D offset from the root, 3D offset from the root) with the
root row holding the absolute root (2D position, 3D position); joints whose predicted score is not positive
stay zero.  `gt` is the ground-truth 3D offset from the root for every joint (root row zero)."""
import json

import numpy as np
import torch
from torch.utils.data.dataset import Dataset


class P2PDataset(Dataset):
    def __init__(self, stage="train", dataset_path="", root_idx=2):
        self.root_idx = root_idx
        with open(dataset_path, "r") as f:
            self.dataset = json.load(f)["3d_pairs"]

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        pair = self.dataset[index]
        p3 = np.asarray(pair["pred_3d"], dtype=np.float64)
        p2 = np.asarray(pair["pred_2d"], dtype=np.float64)
        g3 = np.asarray(pair["gt_3d"], dtype=np.float64)
        r = self.root_idx
        inp = np.zeros((15, 5), np.float64)
        live = p3[:, 3] > 0
        live[r] = False
        inp[live, :2] = p2[live, :2] - p2[r, :2]
        inp[live, 2:] = p3[live, :3] - p3[r, :3]
        inp[r, :2], inp[r, 2:] = p2[r, :2], p3[r, :3]
        gt = g3 - g3[r]
        return torch.from_numpy(inp.reshape(-1)).float(), torch.from_numpy(gt.reshape(-1)).float()
