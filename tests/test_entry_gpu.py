"""GPU end-to-end: the reference's command line (`python test.py -t run_inference ...`) on a
tiny image folder with a recipe checkpoint.  The JSON it writes must equal, record for record,
what the CPU oracle makes of the SAME network output (association is bit-exact given
identical heat-maps; the backbone itself is covered in test_backbone_gpu.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import make_cfg
from recipe import recipe_state_dict
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_inference_cli_end_to_end(tmp_path):
    from model.smap import SMAP
    from model.refinenet import RefineNet
    from dataset.custom_dataset import CustomDataset
    from exps.stage3_root2.config import cfg
    from exps.stage3_root2.test_util import default_cams, merge_flip
    imgdir = tmp_path / "imgs"
    imgdir.mkdir()
    rng = np.random.default_rng(5)
    for i, (h, w) in enumerate([(512, 832), (480, 640), (1080, 1920)]):
        np.save(imgdir / f"f{i}.npy", rng.integers(0, 255, (h, w, 3), dtype=np.uint8))
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = recipe_state_dict(net.state_dict())
    # bias the key-point heads so that peaks (incl. pelvis) exist: scores must exceed 0.2*255
    for k in list(sd):
        if k.endswith("up4.res_conv2.bn.bias"):
            sd[k] = sd[k] + 40.0
    net.load_state_dict(sd)
    rnet = RefineNet().eval()
    rsd = recipe_state_dict(rnet.state_dict())
    rnet.load_state_dict(rsd)
    torch.save({"model": sd}, tmp_path / "SMAP.pth")
    torch.save(rsd, tmp_path / "RefineNet.pth")
    env = dict(os.environ, PROJECT_HOME=str(tmp_path), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "exps", "stage3_root2", "test.py"),
                        "-p", str(tmp_path / "SMAP.pth"), "-rp", str(tmp_path / "RefineNet.pth"), "-t", "run_inference",
                        "-d", "test", "--batch_size", "2", "--do_flip", "1", "--dataset_path", str(imgdir),
                        "--json_name", "e2e"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    out = tmp_path / "model_logs" / "stage3_root2" / "result" / "stage3_root2_run_inference_test_e2e.json"
    res = json.loads(out.read_text())
    assert res["model_pattern"] == "MIX" and isinstance(res["3d_pairs"], list)
    # the same pipeline in-process up to the network output, then the CPU oracle
    ds = CustomDataset(cfg, str(imgdir))
    dev = "cuda:0"
    net = net.to(dev)
    from smap_amd.model.refinenet import RefineNet as RN
    wt, bs = rnet.folded("cpu")
    W = [w.t().contiguous().numpy() for w in wt]
    Bs = [b.numpy() for b in bs]
    expect = []
    for st in range(0, len(ds), 2):
        items = [ds[i] for i in range(st, min(st + 2, len(ds)))]
        imgs = torch.stack([it[0] for it in items]).to(dev)
        scales = {k: torch.tensor([it[2][k] for it in items], dtype=torch.float64) for k in items[0][2]}   # default_collate: python float -> f64
        h, d, rd = net(imgs)
        hf, _, _ = net(torch.flip(imgs, [-1]))
        merge_flip(h, hf, cfg)
        h = h.cpu()
        h[:, :15] /= 255
        h[:, 15:] /= 127
        cams = default_cams(scales, len(items))
        for i, it in enumerate(items):
            bodys, _, _ = O.connect(h[i].numpy(), rd[i, 0].cpu().numpy())
            if len(bodys) == 0:
                continue
            p2, p3, rz = O.lift(bodys, d[i].cpu().numpy(), rd[i, 0].cpu().numpy(), cams[i])
            ref = O.refine(p2, p3, W, Bs)
            expect.append(dict(pred_2d=p2.tolist(), pred_3d=ref.tolist(), root_d=rz.tolist(), image_path=it[1],
                               gt_3d=[], gt_2d=[]))
    assert len(expect) >= 1, "test setup must produce at least one frame with persons"
    assert len(res["3d_pairs"]) == len(expect)
    for got, want in zip(res["3d_pairs"], expect):
        assert set(got) == {"pred_2d", "pred_3d", "root_d", "image_path", "gt_3d", "gt_2d"}
        assert got["image_path"] == want["image_path"]
        assert got["pred_2d"] == want["pred_2d"] and got["root_d"] == want["root_d"]       # bit-exact
        assert np.abs(np.asarray(got["pred_3d"]) - np.asarray(want["pred_3d"])).max() < 1e-3 * 100   # 1e-3 m in cm


def test_two_stream_pipeline_equals_serial_path():
    """smap_amd/pipeline.py (post-processing of batch k overlapped with the backbone of batch k+1,
    double-buffered outputs, pinned D2H) returns exactly what the serial calls return."""
    from model.smap import SMAP
    from smap_amd.pipeline import PosePipeline
    from exps.stage3_root2.config import cfg
    from exps.stage3_root2.test_util import poses_from_outputs
    dev = "cuda:0"
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    sd = recipe_state_dict(net.state_dict())
    for k in list(sd):
        if k.endswith("up4.res_conv2.bn.bias"):
            sd[k] = sd[k] + 60.0
    net.load_state_dict(sd)
    net = net.to(dev)
    B = 2
    g = torch.Generator().manual_seed(3)
    batches = [torch.randn(B, 3, 64, 96, generator=g).to(dev) for _ in range(4)]
    cams = np.tile(np.array([0.5, 192, 128, 96, 64, 192, 192, 96, 64], np.float64), (B, 1))
    from exps.stage3_root2.test_util import merge_flip
    want_all, want_flip, n_people = [], [], 0
    for i, x in enumerate(batches):
        for flip, acc in ((False, want_all), (True, want_flip)):
            h, d, rd = net(x)
            if flip:
                hf, _, _ = net(torch.flip(x, [-1]))
                merge_flip(h, hf, cfg)
            p2, p3, rz, counts = poses_from_outputs(h, d, rd, cams, cfg)
            acc += [(f"b{i}/{j}", p2[j, :c].tolist(), p3[j, :c].tolist(), rz[j, :c].tolist())
                    for j, c in enumerate(counts) if c > 0]
            n_people += int(counts.sum())
    assert n_people > 0
    for depth, flip in ((1, False), (2, False), (3, False), (2, True)):
        pipe = PosePipeline(net, cfg, B, 64, 96, dev, depth=depth, do_flip=flip)
        got = []
        for rep in range(2):                                  # reuse of slots / arenas across many submits
            for i, x in enumerate(batches):
                r = pipe.submit(x, cams, [f"b{i}/{j}" for j in range(B)])
                if r is not None:
                    got += r
        got += pipe.flush() or []
        have = [(r["image_path"], r["pred_2d"], r["pred_3d"], r["root_d"]) for r in got]
        want = want_flip if flip else want_all
        assert have == want + want, (depth, flip)             # in order, bit for bit


def test_device_preprocess_equals_host_dataset(tmp_path):
    """smap_preprocess (HIP) == dataset/custom_dataset.py (host: torch bilinear, round, pad 128, normalise),
    bit for bit, for wide / tall / exact / tiny / odd-sized images."""
    from dataset.custom_dataset import CustomDataset
    from exps.stage3_root2.config import cfg
    from smap_amd.preprocess import preprocess_batch
    rng = np.random.default_rng(11)
    sizes = [(300, 1000), (900, 400), (512, 832), (37, 53), (1080, 1920), (511, 833), (2000, 3001)]
    for i, (h, w) in enumerate(sizes):
        np.save(tmp_path / f"im{i:02d}.npy", rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    ds = CustomDataset(cfg, str(tmp_path))
    raws = [ds.raw(i)[0] for i in range(len(ds))]
    got, scales = preprocess_batch(raws, cfg.INPUT.MEANS, cfg.INPUT.STDS, torch.device("cuda:0"))
    torch.cuda.synchronize()
    for i in range(len(ds)):
        want, name, scale = ds[i]
        assert torch.equal(got[i].cpu(), want), (name, (got[i].cpu() - want).abs().max().item())
        for k in scale:
            assert scales[k][i] == scale[k]
