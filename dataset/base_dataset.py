"""Annotated dataset for the ground-truth modes of test.py (reference: dataset/base_dataset.py:17-171).

Only the inference-side stages exist here -- `test` (the validation frames of cfg.TEST.JSON_PATH, images under
cfg.TEST.ROOT_PATH) and `generation` (the training frames of the 3D datasets, the input of RefineNet training-pair
export); stage `train` builds heat-map / PAF labels for training, which is outside this repository.

A sample is what the reference returns for these stages (base_dataset.py:157-166):
    (normalised BGR CHW image, annotations [MAX_PEOPLE,15,C] fp32 zero padded, image path, scale dict)
produced by the no-augmentation path of aug_croppad (ImageAugmentation.py:54-111): resize by
scale = min(832/w, 512/h), paste centred on a 128-grey 832x512 canvas (the window is centred on
int(image centre * scale)), and move the annotations by the same similarity; joints that leave the canvas get
score 0 (ImageAugmentation.py:103-109, base_dataset.py:112-122).

Annotation JSON: {"root": [{"dataset", "img_paths", "img_width", "img_height", "isValidation",
"bodys": [P][15][C] with C = 11: x, y, Z, visibility(0 none, 1 occluded, 2 visible), X, Y, Z, f_x, f_y, cx, cy}]}.
cv2 is not part of this image: decode and resize are the ones of dataset/custom_dataset.py (PIL -> BGR, fp32
bilinear with half-pixel centres), which can differ from OpenCV's fixed-point resize by 1 LSB."""
import copy
import json
import os.path as osp

import numpy as np
import torch
from torch.utils.data import Dataset

from dataset.custom_dataset import CustomDataset


def croppad_geometry(img_w, img_h, crop_x, crop_y):
    """(scale, (nh, nw), (left, top)): resized size and the integer offset of the resized image inside the
    crop_x x crop_y canvas, following ImageAugmentation.py:63-99 with no augmentation."""
    scale = min(crop_x / float(img_w), crop_y / float(img_h))
    nw, nh = int(round(img_w * scale)), int(round(img_h * scale))            # cv2.resize(fx, fy): dsize = round(src * f)
    center = (np.array([img_w // 2, img_h // 2]) * scale + np.array([0, 0])).astype(int)
    offset = np.array([crop_x / 2 - center[0], crop_y / 2 - center[1]]).astype(int)
    return scale, (nh, nw), (int(offset[0]), int(offset[1]))


class JointDataset(Dataset):
    def __init__(self, cfg, stage, transform=None, with_augmentation=False, with_mds=False):
        if stage == "train":
            raise NotImplementedError("stage 'train' (label generation for training) is outside this repository")
        assert stage in ("test", "generation")
        if with_augmentation:
            raise NotImplementedError("augmentation belongs to training")
        self.stage = stage
        ds = cfg.dataset
        if stage == "generation":
            data = []
            for name in ds.USED_3D_DATASETS:                                   # base_dataset.py:45-50
                with open(ds["%s_JSON_PATH" % name]) as f:
                    data = json.load(f)["root"] + data
        else:
            with open(cfg.TEST.JSON_PATH) as f:
                data = json.load(f)["root"]
        val = [d for d in data if d["isValidation"] != 0]
        trn = [d for d in data if d["isValidation"] == 0]
        self.data = trn if stage == "generation" else val                      # base_dataset.py:86-90
        self.test_root_path = cfg.TEST.ROOT_PATH
        self.root_path = {name: ds["%s_ROOT_PATH" % name] for name in ["COCO"] + list(ds.USED_3D_DATASETS)
                          if ("%s_ROOT_PATH" % name) in ds}
        self.crop_y, self.crop_x = ds.INPUT_SHAPE
        self.keypoint_num = ds.KEYPOINT.NUM
        self.max_people = cfg.DATASET.MAX_PEOPLE
        self.mean = torch.tensor(cfg.INPUT.MEANS, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(cfg.INPUT.STDS, dtype=torch.float32).view(3, 1, 1)

    def __len__(self):
        return len(self.data)

    def image_path(self, d):
        root = self.test_root_path if self.stage == "test" else self.root_path[d["dataset"].upper()]
        return osp.join(root, d["img_paths"])

    def annotate(self, d):
        """(bodys [P,15,C] float64 moved into the network frame, scale, geometry) of one JSON entry."""
        bodys = np.asarray(copy.deepcopy(d["bodys"]), dtype=np.float64)
        scale, size, (left, top) = croppad_geometry(int(d["img_width"]), int(d["img_height"]), self.crop_x, self.crop_y)
        bodys[:, :, :2] *= scale
        bodys[:, :, :2] += np.array([left, top])
        out = (bodys[:, :, 0] >= self.crop_x) | (bodys[:, :, 0] < 0) | (bodys[:, :, 1] >= self.crop_y) | (bodys[:, :, 1] < 0)
        bodys[:, :, 3][out] = 0
        return bodys, scale, size, (left, top)

    def __getitem__(self, index):
        from smap_amd.preprocess import resize_linear_u8
        d = self.data[index]
        bodys, scale, (nh, nw), (left, top) = self.annotate(d)
        img = CustomDataset._read_bgr(self.image_path(d))
        r = resize_linear_u8(img, nh, nw, fx=scale, fy=scale)       # cv2.resize(img, (0, 0), fx=scale, fy=scale): ImageAugmentation.py:70
        canvas = np.full((self.crop_y, self.crop_x, 3), 128, np.uint8)
        # intersection of the resized image (placed at (left, top)) with the canvas
        x0, y0 = max(left, 0), max(top, 0)
        x1, y1 = min(left + nw, self.crop_x), min(top + nh, self.crop_y)
        if x1 > x0 and y1 > y0:
            canvas[y0:y1, x0:x1] = r[y0 - top:y1 - top, x0 - left:x1 - left]
        t = torch.from_numpy(canvas).permute(2, 0, 1).float().div(255.0)
        padded = np.zeros((self.max_people, self.keypoint_num, bodys.shape[2]), np.float64)
        padded[:len(bodys)] = bodys
        meta = {"scale": scale, "img_width": int(d["img_width"]), "img_height": int(d["img_height"]),
                "net_width": self.crop_x, "net_height": self.crop_y}
        return (t - self.mean) / self.std, torch.from_numpy(padded).float(), d["img_paths"], meta
