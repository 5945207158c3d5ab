"""Image-folder dataset for `run_inference` (reference: dataset/custom_dataset.py:11-68).

Same outputs: (normalised BGR CHW float tensor letter-boxed to 832x512 with 128-grey padding,
image name, `scale` dict with scale/img_width/img_height/net_width/net_height).
cv2 and torchvision are not part of this image: decoding uses PIL (converted to BGR like
cv2.imread) and the resize restates OpenCV's 8-bit INTER_LINEAR from its published algorithm, operation by operation
(smap_amd.preprocess.resize_linear_u8: 11-bit fixed-point coefficients, (d + 0.5) / f - 0.5 source coordinates, the integer
vertical pass) -- faithful to the algorithm, unpinned by execution (no cv2 here to run).
`.npy` files holding an HxWx3 uint8 BGR array are accepted as well.
Record order: the reference lists jpg, then png, then jpeg files in glob (= directory) order, which is not defined; here
each extension's files are SORTED (one of the orders the reference may produce, and a reproducible one).  Decoding honours
the EXIF orientation tag like cv2.imread(IMREAD_COLOR) does.
"""
import glob
import os.path as osp

import numpy as np
import torch
from torch.utils.data import Dataset


class CustomDataset(Dataset):
    def __init__(self, cfg, dataset_path):
        self.dataset_path = dataset_path
        self.image_list = []
        for ext in ("jpg", "png", "jpeg", "npy"):
            self.image_list.extend(sorted(glob.glob(osp.join(dataset_path, f"**/*.{ext}"), recursive=True)))
        self.list_size = len(self.image_list)
        self.net_input_shape = (832, 512)       # (width, height)
        self.mean = torch.tensor(cfg.INPUT.MEANS, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(cfg.INPUT.STDS, dtype=torch.float32).view(3, 1, 1)

    def __len__(self):
        return self.list_size

    @staticmethod
    def _read_bgr(path):
        from dataset.decode import read_bgr          # (numpy + PIL only: the decode workers of the CLI import it without torch)
        return read_bgr(path)

    def __getitem__(self, index):
        image_path = self.image_list[index].rstrip()
        image_name = image_path.replace(self.dataset_path, "").lstrip("/")
        img = self._read_bgr(image_path)
        self.image_shape = (img.shape[1], img.shape[0])
        net_img, scale = self.aug_croppad(img)
        t = torch.from_numpy(net_img).permute(2, 0, 1).float().div(255.0)      # ToTensor
        return (t - self.mean) / self.std, image_name, scale                    # Normalize

    def aug_croppad(self, img):
        from smap_amd.preprocess import letterbox_geometry, resize_linear_u8
        crop_x, crop_y = self.net_input_shape
        w0, h0 = self.image_shape
        scale, (nh, nw, top, left) = letterbox_geometry(w0, h0, crop_x, crop_y)   # cv2.resize(fx, fy): dsize = round(src * f)
        r = resize_linear_u8(img, nh, nw, fx=scale["scale"], fy=scale["scale"])     # cv2.resize(img, (0, 0), fx=s, fy=s)
        out = np.full((crop_y, crop_x, 3), 128, np.uint8)
        hh, ww = min(nh, crop_y - top), min(nw, crop_x - left)
        out[top:top + hh, left:left + ww] = r[:hh, :ww]
        return out, scale

    def raw(self, index):
        """(uint8 HxWx3 BGR image, image name) for the device pre-processing path (smap_amd/preprocess.py)."""
        image_path = self.image_list[index].rstrip()
        return self._read_bgr(image_path), image_path.replace(self.dataset_path, "").lstrip("/")
