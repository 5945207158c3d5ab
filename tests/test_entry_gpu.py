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


@pytest.fixture(autouse=True)
def _summation_order_independent_of_the_batch(monkeypatch):
    """Several tests of this file compare schedules of DIFFERENT batch sizes bit for bit (a 4-frame flip schedule against two 2-frame
    forwards, coalesced 6-frame launches against 2-frame ones, the CLI against in-process calls).  That needs the same summation order
    per layer on both sides: they already take the batch-independent heuristic tiles instead of the measured table, and they switch
    split K off (its number of K parts follows the number of output tiles, i.e. the batch; tests/test_backbone_gpu.py covers it) and name
    the whole-block launches explicitly (by default a schedule of <= 2 full-size frames runs layer2's blocks as three launches and layer1's
    on 4 x 16 pixel tiles)."""
    monkeypatch.setenv("SMAP_SPLITK", "0")
    monkeypatch.setenv("SMAP_BLOCK", "64:91,128:94")
    monkeypatch.setenv("SMAP_BLOCK_FIRST", "64:93")


def test_run_inference_cli_end_to_end(tmp_path, monkeypatch):
    # The CLI runs the flip-TTA inside ONE 4-frame schedule, the check below runs two 2-frame forwards: bit-equal results need the
    # same kernel family per layer on both sides (halo / im2col / persistent kernels sum in different orders), so both sides
    # take the batch-size-independent heuristic tiles instead of the measured table (whose entries are per batch size).
    import smap_amd.engine as E
    monkeypatch.setattr(E, "_TILE_TABLE_X3", {})
    monkeypatch.setattr(E, "_TILE_TABLE", {})
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
    env = dict(os.environ, PROJECT_HOME=str(tmp_path), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               SMAP_NO_TILE_TABLE="1")
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


def test_bench_main_two_ranks_on_one_gpu():
    """The REAL multi-rank branch of bench.py (init, per-rank pinning, timed loop, ONE end-of-run gather on the comm stream,
    barrier, MAX over ranks, per-rank host CPU gather, one JSON line on rank 0) under torch.distributed.run with two ranks.
    A one-GPU box cannot host two RCCL ranks, so both share cuda:0 and the collectives use gloo (bench.py's two test
    hooks); everything else is the code the driver's 8-GPU run executes."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SMAP_BENCH_SHARE_GPU="1", SMAP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "2"], capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["frames_per_step"] == 16
    assert len(d["config"]["host_ms_per_step"]["process_cpu_per_rank"]) == 2
    # every rank's records of its 4 timed batches arrived in the gather: >= 1 record per frame of the K != 0 scenes and of
    # the network output (the calibrated heads give ~20 skeletons per frame)
    assert d["config"]["records_in_run"] >= 2 * 4 * 8
    assert "cpu_baseline" not in d                       # reported at N = 1 only


def test_rccl_gather_runs_on_hardware_with_one_rank():
    """RCCL itself, on a one-GPU box: bench.py's REAL `nccl` branch (process group on the device, payload on the device, the end-of-run
    gather on the comm stream inside the timed region, barrier) with ONE rank -- SMAP_FORCE_GATHER=1 skips the world == 1 shortcut
    of smap_amd/dist.py::gather_bytes, so the records really go through dist.all_gather on device tensors
    (lib/utils/comm.py:47-87) and must come back byte for byte.  Then the helper alone on payloads of ragged sizes."""
    env = dict(os.environ, SMAP_FORCE_GATHER="1", SMAP_BENCH_NO_LF0="1")
    env.pop("SMAP_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert c["ranks_in_gather"] == 1 and c["gather"]["backend"] == "nccl" and c["gather"]["payload_device"].startswith("cuda")
    assert c["gather"]["own_payload_returned_identical"] is True and c["gather"]["payload_bytes"] > 10000
    assert c["records_in_run"] >= 4 * 8 and c["timed_steps_reproduce"]["identical_records_per_frame_across_steps"]
    code = (
        "import os, pickle, torch, torch.distributed as dist\n"
        "os.environ['SMAP_FORCE_GATHER'] = '1'\n"
        "from smap_amd.dist import init_single_rank_group, gather_bytes, gather_records\n"
        "torch.cuda.set_device(0)\n"
        "init_single_rank_group('nccl', torch.device('cuda:0'))\n"
        "assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "for n in (0, 1, 7, 4096, 1 << 20):\n"
        "    p = bytes(bytearray((i * 131 + n) % 251 for i in range(n)))\n"
        "    got = gather_bytes(p, 'cuda:0')\n"
        "    assert got == [p], n\n"
        "recs = [{'pred_3d': [[1.5, 2.5]], 'image_path': 'a/b'}]\n"
        "assert gather_records(recs, 'cuda:0') == [recs]\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-3000:]


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
    # chunk = frames per backbone launch: 1 < batch exercises the split a 4 GiB arena forces on the shipped test.sh settings
    for depth, flip, chunk in ((1, False, None), (2, False, None), (3, False, None), (2, True, None), (2, False, 1), (2, True, 1)):
        pipe = PosePipeline(net, cfg, B, 64, 96, dev, depth=depth, do_flip=flip, max_frames_per_launch=chunk)
        assert pipe.chunk == (chunk or B)
        got = []
        for rep in range(2):                                  # reuse of slots / arenas across many submits
            for i, x in enumerate(batches):
                r = pipe.submit(x, cams, [f"b{i}/{j}" for j in range(B)])
                if r is not None:
                    got += r
        got += pipe.flush() or []
        have = [(r["image_path"], r["pred_2d"], r["pred_3d"], r["root_d"]) for r in got]
        want = want_flip if flip else want_all
        assert have == want + want, (depth, flip, chunk)      # in order, bit for bit
    # coalesced launches (make_pipeline): `group` submitted batches per backbone launch; 8 submits with group 3 leave an
    # incomplete group for flush() -- the same records, in the same order, from 6-frame launches + two 2-frame ones
    from smap_amd.pipeline import CoalescedPipeline, make_pipeline
    assert not isinstance(make_pipeline(net, cfg, B, 64, 96, dev, launch_frames=0), CoalescedPipeline)
    for launch_frames, flip in ((4, False), (6, False), (8, True), (16, False)):
        pipe = make_pipeline(net, cfg, B, 64, 96, dev, launch_frames=launch_frames, depth=2, do_flip=flip)
        assert isinstance(pipe, CoalescedPipeline) and pipe.B == B and pipe.frames_per_launch == pipe.group * B
        assert pipe.group == launch_frames // (B * (2 if flip else 1))
        got = []
        for rep in range(2):
            for i, x in enumerate(batches):
                got += pipe.submit(x, cams, [f"b{i}/{j}" for j in range(B)]) or []
        got += pipe.flush() or []
        have = [(r["image_path"], r["pred_2d"], r["pred_3d"], r["root_d"]) for r in got]
        want = want_flip if flip else want_all
        assert have == want + want, (launch_frames, flip)
    # `extra` maps (bench only) through coalesced launches: the caller's per-batch tensors are concatenated into temporaries the
    # post stream must keep alive, and every frame keeps the prefix of ITS batch.  Extra = the network's own scaled maps of
    # that batch, so "again<i>/<tag>" must carry exactly the record of <tag>.
    from smap_amd.dapalib import scale_hms_
    ex = []
    for x in batches:
        h, d, rd = net(x)
        h = scale_hms_(h.clone().contiguous())          # the pipeline's own scaling kernel (test.py:111-112)
        ex.append((h, rd.clone(), d.clone()))
    pipe = make_pipeline(net, cfg, B, 64, 96, dev, launch_frames=6, depth=2, n_extra=1)
    got = []
    for i, x in enumerate(batches):
        got += pipe.submit(x, cams, [f"b{i}/{j}" for j in range(B)], extra=[(f"again{i}", *ex[i])]) or []
    got += pipe.flush() or []
    by = {r["image_path"]: (r["pred_2d"], r["pred_3d"], r["root_d"]) for r in got}
    assert len(by) == len(got) == 2 * len(want_all)
    for name, p2, p3, rz in want_all:
        i = name[1:name.index("/")]
        assert by[name] == (p2, p3, rz), name
        assert by[f"again{i}/{name}"] == (p2, p3, rz), "again/" + name


@pytest.mark.parametrize("flip,chunk", [(False, None), (True, None), (False, 2)])
def test_one_overflowing_frame_does_not_condemn_its_launch(flip, chunk):
    """A frame whose activations leave the fp16 range ends as NaN maps (status word).  The reference's fp32 loop would carry on
    with the other frames; so does the pipeline: that frame has no record (RuntimeWarning, `dropped_frames`), the other frames
    of the SAME launch keep theirs, bit for bit; strict_nonfinite=True raises instead."""
    from model.smap import SMAP
    from smap_amd.pipeline import PosePipeline
    from exps.stage3_root2.config import cfg
    dev = "cuda:0"
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    sd = recipe_state_dict(net.state_dict())
    for k in list(sd):
        if k.endswith("up4.res_conv2.bn.bias"):
            sd[k] = sd[k] + 60.0
    net.load_state_dict(sd)
    net = net.to(dev)
    B = 4
    x = torch.randn(B, 3, 64, 96, generator=torch.Generator().manual_seed(3)).to(dev)
    bad = x.clone()
    bad[2] *= 1e5                                               # frame 2 overflows (test_split_precision_dynamic_range: scale 1e5)
    cams = np.tile(np.array([0.5, 192, 128, 96, 64, 192, 192, 96, 64], np.float64), (B, 1))
    tags = [f"f{j}" for j in range(B)]
    key = lambda recs: [(r["image_path"], r["pred_2d"], r["pred_3d"], r["root_d"]) for r in recs]
    pipe = PosePipeline(net, cfg, B, 64, 96, dev, depth=2, do_flip=flip, max_frames_per_launch=chunk)
    pipe.submit(x, cams, tags)
    clean = pipe.flush()
    assert {r["image_path"] for r in clean} == set(tags) and not pipe.dropped_frames
    pipe.submit(bad, cams, tags)
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        got = pipe.flush()
    assert pipe.dropped_frames == ["f2"]
    assert key(got) == key([r for r in clean if r["image_path"] != "f2"])
    strict = PosePipeline(net, cfg, B, 64, 96, dev, depth=1, do_flip=flip, max_frames_per_launch=chunk, strict_nonfinite=True)
    strict.submit(bad, cams, tags)
    with pytest.raises(RuntimeError, match="f2"):
        strict.flush()
    if not flip and chunk is None:      # the same through the front end test.py gets for small batches (make_pipeline -> CoalescedPipeline)
        from smap_amd.pipeline import CoalescedPipeline, make_pipeline
        co = make_pipeline(net, cfg, 2, 64, 96, dev, launch_frames=4)
        assert isinstance(co, CoalescedPipeline) and co.dropped_frames == [] and co.strict_nonfinite is False
        co.submit(bad[:2], cams[:2], tags[:2])
        co.submit(bad[2:], cams[2:], tags[2:])
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            got2 = co.flush()
        assert co.dropped_frames == ["f2"] and {r["image_path"] for r in got2} == set(tags) - {"f2"}
        co.strict_nonfinite = True
        assert co.inner.strict_nonfinite and co._small.strict_nonfinite


def test_arena_budget_splits_a_batch_into_smaller_launches(monkeypatch):
    """SMAP_MAX_ARENA_BYTES (default: 45 % of the free device memory) bounds ONE activation arena; a batch whose schedule would exceed
    it runs as several smaller launches (PosePipeline's split loop), same records."""
    from model.smap import SMAP
    from smap_amd.engine import ArenaTooLarge, Graph
    from smap_amd.pipeline import PosePipeline
    from exps.stage3_root2.config import cfg
    dev = "cuda:0"
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    sd = recipe_state_dict(net.state_dict())
    for k in list(sd):
        if k.endswith("up4.res_conv2.bn.bias"):
            sd[k] = sd[k] + 60.0
    net.load_state_dict(sd)
    net = net.to(dev)
    B = 4
    x = torch.randn(B, 3, 64, 96, generator=torch.Generator().manual_seed(3)).to(dev)
    cams = np.tile(np.array([0.5, 192, 128, 96, 64, 192, 192, 96, 64], np.float64), (B, 1))
    tags = [f"f{j}" for j in range(B)]
    key = lambda recs: [(r["image_path"], r["pred_2d"], r["root_d"]) for r in recs]
    whole = PosePipeline(net, cfg, B, 64, 96, dev)
    assert whole.chunk == B
    whole.submit(x, cams, tags)
    want = whole.flush()
    need = {b: Graph(sd, b, 64, 96, precision=net.precision).allocate() for b in (1, 2, 4)}
    assert need[1] < need[2] < need[4]
    net.invalidate_engine()
    monkeypatch.setenv("SMAP_MAX_ARENA_BYTES", str(need[2] + 4096))
    split = PosePipeline(net, cfg, B, 64, 96, dev)
    assert split.chunk == 2 and split.engine.graph.arena_bytes <= need[2] + 4096
    split.submit(x, cams, tags)
    got = split.flush()
    assert [k_[0] for k_ in key(got)] == [k_[0] for k_ in key(want)] and len(got) == len(want)
    monkeypatch.setenv("SMAP_MAX_ARENA_BYTES", str(need[1] - 1))
    net.invalidate_engine()
    with pytest.raises(ArenaTooLarge, match="budget"):
        PosePipeline(net, cfg, B, 64, 96, dev)


def test_status_words_cover_launches_beyond_31_frames():
    """One status bit per output frame, SMAP_STATUS_WORDS(B) words (include/smap_hip.h): in a 40-frame launch the overflow of frame 33
    must not be booked on frame 2 (round 4 kept ONE word and folded frames modulo 31), and the pipeline drops exactly that frame."""
    from model.smap import SMAP
    from smap_amd.pipeline import PosePipeline
    from exps.stage3_root2.config import cfg
    dev = "cuda:0"
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    sd = recipe_state_dict(net.state_dict())
    for k in list(sd):
        if k.endswith("up4.res_conv2.bn.bias"):
            sd[k] = sd[k] + 60.0
    net.load_state_dict(sd)
    net = net.to(dev)
    B = 40
    x = torch.randn(B, 3, 64, 96, generator=torch.Generator().manual_seed(3)).to(dev)
    x[33] *= 1e5
    x[31] *= 1e5
    eng = net.engine(B, 64, 96, torch.device(dev))
    assert eng.status_words == 2
    out = eng.new_output()
    hms, _, _ = eng.run(x, out=out)
    torch.cuda.synchronize()
    assert eng.status(out) & 1 and eng.bad_frames(out) == [31, 33]
    assert [f for f in range(B) if not torch.isfinite(hms[f]).all()] == [31, 33]
    cams = np.tile(np.array([0.5, 192, 128, 96, 64, 192, 192, 96, 64], np.float64), (B, 1))
    pipe = PosePipeline(net, cfg, B, 64, 96, dev, depth=1)
    pipe.submit(x, cams, [f"f{j}" for j in range(B)])
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        got = pipe.flush()
    assert pipe.dropped_frames == ["f31", "f33"] and {r["image_path"] for r in got} == {f"f{j}" for j in range(B)} - {"f31", "f33"}


def test_device_preprocess_equals_host_dataset(tmp_path):
    """smap_preprocess (HIP) == dataset/custom_dataset.py (host: OpenCV's fixed-point INTER_LINEAR restated in numpy, pad 128,
    normalise), bit for bit, for wide / tall / exact / tiny / odd-sized images and an exact 2x shrink (box-mean path)."""
    from dataset.custom_dataset import CustomDataset
    from exps.stage3_root2.config import cfg
    from smap_amd.preprocess import preprocess_batch
    rng = np.random.default_rng(11)
    sizes = [(300, 1000), (900, 400), (512, 832), (37, 53), (1080, 1920), (511, 833), (2000, 3001), (1024, 1664)]
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


def test_cli_device_preprocess_loader_equals_host_loader(tmp_path):
    """`test.py --device_preprocess 1` (decode-ahead thread pool or worker processes, page-locked staging, resize / pad / normalise on the GPU)
    writes the same result file as the reference's loader path (DataLoader + host resize), record for record, bit for bit -- the pre-processing kernel
    is bit-exact and the loader keeps the frame order; five images of four sizes, batch 2 (a ragged last batch)."""
    from model.smap import SMAP
    imgdir = tmp_path / "imgs"
    imgdir.mkdir()
    rng = np.random.default_rng(9)
    for i, (h, w) in enumerate([(512, 832), (480, 640), (1080, 1920), (1024, 1664), (300, 900)]):
        np.save(imgdir / f"f{i}.npy", rng.integers(0, 255, (h, w, 3), dtype=np.uint8))
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = recipe_state_dict(net.state_dict())
    for k in list(sd):
        if k.endswith("up4.res_conv2.bn.bias"):
            sd[k] = sd[k] + 40.0
    torch.save({"model": sd}, tmp_path / "SMAP.pth")
    res = {}
    for tag, extra, env_extra in (("host", [], {}), ("device", ["--device_preprocess", "1"], {"SMAP_DECODE_THREADS": "3"}),
                                  ("procs", ["--device_preprocess", "1"], {"SMAP_DECODE_PROCS": "2", "SMAP_DECODE_SLOT_MB": "4"})):   # (1080p does not fit 4 MB)
        env = dict(os.environ, PROJECT_HOME=str(tmp_path), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **env_extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "exps", "stage3_root2", "test.py"), "-p", str(tmp_path / "SMAP.pth"),
                            "-t", "run_inference", "-d", "test", "--batch_size", "2", "--dataset_path", str(imgdir), "--json_name", tag],
                           capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = json.loads((tmp_path / "model_logs" / "stage3_root2" / "result" / f"stage3_root2_run_inference_test_{tag}.json").read_text())
    assert len(res["host"]["3d_pairs"]) >= 3, "the set-up must produce frames with persons"
    assert res["device"] == res["host"]
    assert res["procs"] == res["host"]                # decoders as worker processes over shared memory (SMAP_DECODE_PROCS): the same file


def _annotated_set(tmp_path, net, dev, sizes, seed):
    """A tiny annotated dataset whose ground truth sits near (some of) the persons the recipe network detects:
    images first, then annotations placed by inverting the letter-box of the detected roots."""
    from exps.stage3_root2.config import cfg
    from dataset.base_dataset import JointDataset, croppad_geometry
    rng = np.random.default_rng(seed)
    root = tmp_path / "MultiPersonTestSet"
    entries = []
    for i, (h, w) in enumerate(sizes):
        (root / f"TS{i + 1}").mkdir(parents=True, exist_ok=True)
        np.save(root / f"TS{i + 1}" / f"img_{i:06d}.npy", rng.integers(0, 255, (h, w, 3), dtype=np.uint8))
        entries.append({"dataset": "MUCO", "img_paths": f"TS{i + 1}/img_{i:06d}.npy", "img_width": w, "img_height": h,
                        "isValidation": 1 if i % 3 else 0, "bodys": np.zeros((1, 15, 11)).tolist()})
    (root / "M3E_gt.json").write_text(json.dumps({"root": entries}))
    cfg.TEST.ROOT_PATH, cfg.TEST.JSON_PATH = str(root), str(root / "M3E_gt.json")
    for e in entries:                                                     # detect, then annotate
        h, w = e["img_height"], e["img_width"]
        scale, (nh, nw), (left, top) = croppad_geometry(w, h, 832, 512)
        img = np.load(root / e["img_paths"])
        from smap_amd.preprocess import resize_linear_u8
        canvas = np.full((512, 832, 3), 128, np.uint8)
        r = resize_linear_u8(img, nh, nw, fx=scale, fy=scale)
        x0, y0, x1, y1 = max(left, 0), max(top, 0), min(left + nw, 832), min(top + nh, 512)
        canvas[y0:y1, x0:x1] = r[y0 - top:y1 - top, x0 - left:x1 - left]
        t = torch.from_numpy(canvas).permute(2, 0, 1).float().div(255.0)
        t = (t - torch.tensor(cfg.INPUT.MEANS).view(3, 1, 1)) / torch.tensor(cfg.INPUT.STDS).view(3, 1, 1)
        hm, _, rd = net(t[None].to(dev))
        hm = hm.cpu()
        hm[:, :15] /= 255
        hm[:, 15:] /= 127
        bodys, _, _ = O.connect(hm[0].numpy(), rd[0, 0].cpu().numpy())
        P = len(bodys)
        G = max(2, min(P, 5))
        ann = np.zeros((G, 15, 11))
        for g in range(G):
            if g < P and g % 3 != 2:                                    # near a detected root (matched)
                net_xy = bodys[g, 2, :2] * 4 + rng.normal(0, 5, 2)
            else:                                                        # far from everything, or occluded below
                net_xy = np.array([rng.uniform(5, 825), rng.uniform(5, 505)])
            for j in range(15):
                xy = net_xy + (rng.normal(0, 30, 2) if j != 2 else 0)
                ann[g, j, :2] = (xy - np.array([left, top])) / scale
                ann[g, j, 2] = rng.uniform(200, 500)
                ann[g, j, 3] = 2
                ann[g, j, 4:7] = rng.normal(0, 60, 3) + np.array([0, 0, 300])
                ann[g, j, 7:11] = [1500.0, 1490.0, w / 2 + 1.5, h / 2 - 0.5]
        ann[G - 1, 2, 3] = 1                                              # occluded root: dropped (test.py:78)
        e["bodys"] = ann.tolist()
    (root / "M3E_gt.json").write_text(json.dumps({"root": entries}))
    muco = tmp_path / "data" / "MuCo"
    (muco / "annotations").mkdir(parents=True, exist_ok=True)
    for e in entries:                                                     # the same frames double as the "MuCo" set
        (muco / os.path.dirname(e["img_paths"])).mkdir(parents=True, exist_ok=True)
        np.save(muco / e["img_paths"], np.load(root / e["img_paths"]))
    (muco / "annotations" / "MuCo.json").write_text(json.dumps({"root": entries}))
    return root


@pytest.mark.parametrize("mode,data_mode,refine", [("generate_result", "test", True), ("generate_train", "generation", True),
                                                   ("generate_train", "test", False)])
def test_ground_truth_modes_cli_end_to_end(tmp_path, monkeypatch, mode, data_mode, refine):
    """`test.py -t generate_result|generate_train` (test.py:73-95,142-143) on a tiny annotated set: the JSON must
    equal what the CPU oracle (register_gt + f64 lifting + RefineNet) makes of the SAME network output."""
    from model.smap import SMAP
    from model.refinenet import RefineNet
    from exps.stage3_root2.config import cfg
    from dataset.base_dataset import JointDataset
    from smap_amd.records import annotation_camera, frame_record, kept_annotations, train_records
    import smap_amd.engine as E                           # batch 2 (CLI) vs batch 1 (below) must pick the same kernels: heuristic tiles
    monkeypatch.setattr(E, "_TILE_TABLE_X3", {})
    monkeypatch.setattr(E, "_TILE_TABLE", {})
    dev = "cuda:0"
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = recipe_state_dict(net.state_dict())
    for k in list(sd):
        if k.endswith("up4.res_conv2.bn.bias"):
            sd[k] = sd[k] + 40.0
    net.load_state_dict(sd)
    net = net.to(dev)
    rnet = RefineNet().eval()
    rsd = recipe_state_dict(rnet.state_dict())
    rnet.load_state_dict(rsd)
    torch.save({"model": sd}, tmp_path / "SMAP.pth")
    torch.save(rsd, tmp_path / "RefineNet.pth")
    root = _annotated_set(tmp_path, net, dev, [(512, 832), (480, 640), (1080, 1920), (2048, 2048), (600, 800)], seed=11)
    env = dict(os.environ, PROJECT_HOME=str(tmp_path), SMAP_TEST_ROOT=str(root), SMAP_NO_TILE_TABLE="1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, os.path.join(ROOT, "exps", "stage3_root2", "test.py"), "-p", str(tmp_path / "SMAP.pth"),
           "-t", mode, "-d", data_mode, "--batch_size", "2", "--json_name", "gt"]
    if refine:
        cmd += ["-rp", str(tmp_path / "RefineNet.pth")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    out = tmp_path / "model_logs" / "stage3_root2" / "result" / f"stage3_root2_{mode}_{data_mode}_gt.json"
    res = json.loads(out.read_text())
    # expectation: same dataset items, network in-process, CPU oracle after the network
    cfg.dataset.MUCO_ROOT_PATH = str(tmp_path / "data" / "MuCo")
    cfg.dataset.MUCO_JSON_PATH = str(tmp_path / "data" / "MuCo" / "annotations" / "MuCo.json")
    ds = JointDataset(cfg, data_mode)
    assert len(ds) == (3 if data_mode == "test" else 2)                   # isValidation split (base_dataset.py:86-90)
    wt, bs = rnet.folded("cpu")
    W, Bs = [w.t().contiguous().numpy() for w in wt], [b.numpy() for b in bs]
    expect, n_matched = [], 0
    for i in range(len(ds)):
        img, meta, path, scale = ds[i]
        gt = kept_annotations(meta.numpy())
        if len(gt) == 0:
            continue
        hm, d, rd = net(img[None].to(dev))
        hm = hm.cpu()
        hm[:, :15] /= 255
        hm[:, 15:] /= 127
        bodys, _, _ = O.connect(hm[0].numpy(), rd[0, 0].cpu().numpy())
        if len(bodys) == 0:
            continue
        m = O.register_gt(bodys, gt[:, 2, :2])
        p2, p3, rz = O.lift_gt(m, d[0].cpu().numpy(), rd[0, 0].cpu().numpy(), annotation_camera(gt, scale))
        if refine:
            p3 = O.refine_gt(p2, p3, W, Bs)
        n_matched += int((p2[:, 2, 3] != 0).sum())
        if mode == "generate_train":
            expect += train_records(p2, p3, rz, gt)
        else:
            expect.append(frame_record(p2, p3, rz, path, gt))
    assert n_matched >= 2, "test setup must produce matched persons"
    assert len(res["3d_pairs"]) == len(expect) and len(expect) >= 1
    for got, want in zip(res["3d_pairs"], expect):
        assert set(got) == set(want)
        for k in want:
            if k == "pred_3d" and refine:
                assert np.abs(np.asarray(got[k]) - np.asarray(want[k])).max() < 1e-3 * 100     # 1e-3 m, in cm
            else:
                assert got[k] == want[k], k                             # matching, 2D, depths, ground truth: exact
    if mode == "generate_result":                                        # and the evaluation hand-off accepts the file
        from lib.eval.convert import convert
        p3d, p2d = convert(str(out), out_dir=str(tmp_path))
        assert set(p3d) == {r["image_path"] for r in res["3d_pairs"]}
