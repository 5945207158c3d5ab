"""CPU suite for the host-side orchestration: dataset letter-boxing, flip-TTA merge, camera
defaults, contiguous frame sharding and the 2-rank gloo gather of result records."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_custom_dataset_letterbox(tmp_path):
    from dataset.custom_dataset import CustomDataset
    from exps.stage3_root2.config import cfg
    rng = np.random.default_rng(0)
    np.save(tmp_path / "wide.npy", rng.integers(0, 255, (300, 1000, 3), dtype=np.uint8))
    np.save(tmp_path / "tall.npy", rng.integers(0, 255, (900, 400, 3), dtype=np.uint8))
    from PIL import Image
    Image.fromarray(rng.integers(0, 255, (512, 832, 3), dtype=np.uint8)).save(tmp_path / "exact.png")
    ds = CustomDataset(cfg, str(tmp_path))
    assert len(ds) == 3
    for i in range(3):
        img, name, scale = ds[i]
        assert img.shape == (3, 512, 832) and img.dtype == torch.float32
        assert scale["net_width"] == 832 and scale["net_height"] == 512
        assert abs(scale["scale"] - min(832 / scale["img_width"], 512 / scale["img_height"])) < 1e-12
        pad = (torch.tensor(128 / 255.0) - ds.mean) / ds.std              # 128-grey padding, normalised
        if "wide" in name:
            assert torch.allclose(img[:, 0, 0], pad.view(3)) and torch.allclose(img[:, -1, -1], pad.view(3))
        if "tall" in name:
            assert torch.allclose(img[:, 0, 0], pad.view(3)) and torch.allclose(img[:, 256, -1], pad.view(3))


def test_merge_flip_matches_reference_loop():
    from exps.stage3_root2.config import cfg
    from exps.stage3_root2.test_util import merge_flip
    g = torch.Generator().manual_seed(0)
    a = torch.randn(2, 43, 8, 12, generator=g)
    b = torch.randn(2, 43, 8, 12, generator=g)
    want = a.clone()
    fl = torch.flip(b, dims=[-1])                                           # test.py:58
    kpt = 15
    pair = cfg.DATASET.KEYPOINT.FLIP_ORDER + [x + kpt for x in cfg.DATASET.PAF.FLIP_CHANNEL]
    for i in range(len(pair)):                                              # test.py:65-69
        if i >= kpt and (i - kpt) % 2 == 0:
            want[:, i] += fl[:, pair[i]] * -1
        else:
            want[:, i] += fl[:, pair[i]]
    want[:, kpt:] *= 0.5
    got = merge_flip(a.clone(), b, cfg)
    assert torch.equal(got, want)


def test_default_cams():
    from exps.stage3_root2.test_util import default_cams
    scales = {"scale": torch.tensor([0.5, 0.25]), "img_width": torch.tensor([1664, 3328]),
              "img_height": torch.tensor([1024, 1000]), "net_width": torch.tensor([832, 832]),
              "net_height": torch.tensor([512, 512])}
    c = default_cams(scales, 2)
    assert c.shape == (2, 9) and c.dtype == np.float64
    assert list(c[1]) == [0.25, 3328, 1000, 832, 512, 3328, 3328, 1664, 500]


def test_shard_range_is_a_contiguous_partition():
    from smap_amd.dist import shard_range
    for n in (0, 1, 7, 8, 9, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            flat = [i for s, e in spans for i in range(s, e)]
            assert flat == list(range(n))
            assert max(e - s for s, e in spans) <= -(-n // world)


_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from smap_amd.dist import gather_json, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
st, ed = shard_range(11, world, rank)
recs = [{"image_path": f"img{i}", "pred_3d": [[float(i), 0.5 * rank]], "root_d": [1.0 / (i + 1)]} for i in range(st, ed)]
parts = gather_json(recs)
flat = [r for p in parts for r in p]
assert [r["image_path"] for r in flat] == [f"img{i}" for i in range(11)], flat
assert flat[7]["root_d"][0] == 1.0 / 8
if rank == 0:
    print("GATHER_OK", len(flat))
dist.destroy_process_group()
"""


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), ROOT],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GATHER_OK 11" in r.stdout


def test_host_resize_matches_torch_bilinear_within_one_lsb():
    """resize_bilinear_u8 is F.interpolate(bilinear, align_corners=False, antialias=False) up to the
    FMA contraction inside ATen's vectorised CPU kernel: <= 1 LSB on a tiny fraction of pixels."""
    import torch.nn.functional as F
    from smap_amd.preprocess import resize_bilinear_u8
    rng = np.random.default_rng(2)
    for (h, w), (nh, nw) in [((300, 1000), (250, 832)), ((900, 400), (512, 228)), ((37, 53), (512, 733)),
                             ((512, 832), (512, 832))]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = resize_bilinear_u8(img, nh, nw).astype(np.int32)
        t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
        ref = F.interpolate(t, size=(nh, nw), mode="bilinear", align_corners=False, antialias=False)
        ref = ref.round().clamp(0, 255)[0].permute(1, 2, 0).numpy().astype(np.int32)
        diff = np.abs(got - ref)
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, ((h, w), diff.max(), (diff > 0).mean())


def test_derived_skeleton_tables_equal_reference_constants():
    """dataset/data_settings.py derives limbs and mirror tables from the skeleton tree; the values are
    the reference's (dataset/data_settings.py:22,28-34)."""
    from dataset.data_settings import MIX
    assert MIX.KEYPOINT.FLIP_ORDER == [0, 1, 2, 9, 10, 11, 12, 13, 14, 3, 4, 5, 6, 7, 8]
    assert MIX.PAF.VECTOR == [[0, 1], [0, 2], [0, 9], [9, 10], [10, 11], [0, 3], [3, 4], [4, 5],
                              [2, 12], [12, 13], [13, 14], [2, 6], [6, 7], [7, 8]]
    assert MIX.PAF.FLIP_CHANNEL == [0, 1, 2, 3, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7, 8, 9,
                                    22, 23, 24, 25, 26, 27, 16, 17, 18, 19, 20, 21]
    assert (MIX.INPUT_SHAPE, MIX.OUTPUT_SHAPE, MIX.STRIDE, MIX.ROOT_IDX) == ((512, 832), (128, 208), 4, 2)
