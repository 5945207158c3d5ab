"""Reader of the RefineNet training pairs that `test.py -t generate_train` writes (reference:
dataset/p2p_dataset.py:9-40) -- the consumer side of that file format.

TOOLING, not part of the inference hot path (SURVEY.md section 2 lists RefineNet training as out of scope): it exists so that the
`generate_train` record format (smap_amd/records.py::train_records) has a reader that pins it (tests/golden/p2p.npz); nothing
under smap_amd/ or exps/ imports it.

Item i: (inp [75] fp32, gt [45] fp32).  `inp` is 15 x (2D offset from the root, 3D offset from the root) with the
root row holding the absolute root (2D position, 3D position); joints whose predicted score is not positive
stay zero.  `gt` is the ground-truth 3D offset from the root for every joint (root row zero)."""
import json

import numpy as np
import torch
from torch.utils.data.dataset import Dataset

N_JOINTS = 15


def pair_to_vectors(record, root):
    """One `3d_pairs` record of the per-person flavour (smap_amd/records.py::train_records) -> (inp [75], gt [45]) f64."""
    pose3 = np.asarray(record["pred_3d"], np.float64)            # [15,4]  X, Y, Z, score
    pose2 = np.asarray(record["pred_2d"], np.float64)            # [15,4]  x, y, relZ, score
    truth = np.asarray(record["gt_3d"], np.float64)              # [15,3]
    feat = np.concatenate([pose2[:, :2] - pose2[root, :2], pose3[:, :3] - pose3[root, :3]], axis=1)   # [15,5]
    feat[pose3[:, 3] <= 0] = 0.0                                 # undetected joints carry no offset
    feat[root] = np.concatenate([pose2[root, :2], pose3[root, :3]])
    return feat.reshape(N_JOINTS * 5), (truth - truth[root]).reshape(N_JOINTS * 3)


class P2PDataset(Dataset):
    def __init__(self, stage="train", dataset_path="", root_idx=2):
        self.root_idx = root_idx
        with open(dataset_path, "r") as f:
            self.dataset = json.load(f)["3d_pairs"]

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        inp, gt = pair_to_vectors(self.dataset[index], self.root_idx)
        return torch.from_numpy(inp).float(), torch.from_numpy(gt).float()
