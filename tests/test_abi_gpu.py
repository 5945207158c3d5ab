"""GPU: the parts of the C ABI a non-Python host relies on -- the serialised plan run by a plain C program from
include/smap_hip.h alone, several input buffers per launch, and arenas beyond 4 GiB."""
import os
import subprocess

import numpy as np
import pytest
import torch

from helpers import make_cfg
from recipe import recipe_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _small_net():
    from smap_amd.model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((16, 24))).eval()
    net.load_state_dict(recipe_state_dict(net.state_dict()))
    return net


@pytest.mark.parametrize("precision", ["x3", "f16"])
def test_c_host_runs_the_forward_from_a_plan_blob(tmp_path, golden_dir, precision):
    """tests/c/blob_runner.c: plain C, gcc, libamdhip64 + libsmap_hip.so, no Python in the process.  It loads the blob
    BackboneEngine.blob() wrote, sizes its buffers from smap_blob_info, uploads the weight section, runs smap_plan_run and writes
    the output buffer: bit-identical to the Python-hosted engine, and within the small-schedule tolerance of the IMPORTED
    reference model's golden output (tests/golden/backbone_small.npz)."""
    exe = tmp_path / "blob_runner"
    rocm = "/opt/rocm"
    subprocess.run(["gcc", "-O1", "-Wall", f"-I{rocm}/include", f"-I{ROOT}/include", os.path.join(ROOT, "tests", "c", "blob_runner.c"),
                    "-o", str(exe), f"-L{rocm}/lib", "-lamdhip64", f"-L{ROOT}/smap_amd", "-lsmap_hip",
                    f"-Wl,-rpath,{rocm}/lib", f"-Wl,-rpath,{ROOT}/smap_amd"], check=True, capture_output=True, text=True)
    z = np.load(f"{golden_dir}/backbone_small.npz")
    net = _small_net().to(DEV)
    net.precision = precision
    x = torch.from_numpy(z["x"])
    eng = net.engine(2, 64, 96, torch.device(DEV))
    (tmp_path / "plan.blob").write_bytes(eng.blob())
    (tmp_path / "in.f32").write_bytes(x.numpy().tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "plan.blob"), str(tmp_path / "in.f32"), str(tmp_path / "out.f32")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    print(r.stdout.strip())
    got = np.frombuffer((tmp_path / "out.f32").read_bytes(), np.float32)
    out = eng.new_output()
    eng.run(x.to(DEV), out=out)
    torch.cuda.synchronize()
    want = out.cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))      # same plan, same kernels, same bits
    n = 2 * 16 * 24
    tol = 2e-5 if precision == "x3" else 2e-2
    for name, lo, c in (("hms", 0, 43), ("det_d", 43 * n, 14), ("root_d", 57 * n, 1)):
        a, b = got[lo:lo + c * n].reshape(2, c, 16, 24), z[name]
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), name
    assert int(got[-1:].view(np.int32)[0]) == 0                                                          # status word


def test_plan_cache_second_start_loads_the_blob(tmp_path, monkeypatch):
    """SMAP_PLAN_CACHE=<dir>: the first engine of a (checkpoint, shape, arithmetic) builds the schedule and leaves Graph.blob() on disk; the next
    process -- here: a fresh model object -- gets plan and weights back through smap_plan_create_from_blob without packing anything, runs the
    same launches (outputs bit for bit) and starts in a fraction of the time.  Other weights, another batch or a damaged file miss the cache."""
    import time
    from smap_amd import engine as E
    monkeypatch.setenv("SMAP_PLAN_CACHE", str(tmp_path))
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(4)).to(DEV)

    def start(scale=None, B=2):
        net = _small_net()
        if scale is not None:
            sd = net.state_dict()
            sd["top.conv.conv.weight"] = sd["top.conv.conv.weight"] * scale
            net.load_state_dict(sd)
        net = net.to(DEV)
        net.precision = "x3"
        t0 = time.perf_counter()
        eng = net.engine(B, 64, 96, torch.device(DEV), scaled_hms=True)
        return eng, time.perf_counter() - t0
    e1, t1 = start()
    blobs = sorted(p.name for p in tmp_path.iterdir())
    assert not e1.from_cache and len(blobs) == 2 and blobs[0].endswith(".smapplan") and blobs[1].endswith(".smapplan.json")
    want = [t.clone() for t in e1.run(x)]
    e2, t2 = start()
    assert e2.from_cache and e2._graph is None and e2.cache_key == e1.cache_key
    got = e2.run(x)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    assert t2 < 0.5 * t1, (t1, t2)
    assert len(e2.graph.ops) == e2.n_ops == e1.n_ops           # the Python view is still there for whoever asks (built lazily)
    print(f"engine start: {t1:.2f} s built, {t2:.2f} s from the plan cache")
    e3, _ = start(scale=1.0001)                                 # other weights -> other key
    assert not e3.from_cache and e3.cache_key != e1.cache_key
    e4, _ = start(B=1)
    assert not e4.from_cache
    path = tmp_path / (e1.cache_key + ".smapplan")
    raw = bytearray(path.read_bytes())
    raw[0:4] = b"XXXX"                                          # damaged: refused by smap_plan_create_from_blob, rebuilt, rewritten
    path.write_bytes(bytes(raw))
    e5, _ = start()
    assert not e5.from_cache and all(torch.equal(a, b) for a, b in zip(e5.run(x), want))
    e6, _ = start()
    assert e6.from_cache
    assert len([p for p in tmp_path.iterdir() if p.name.endswith(".smapplan")]) <= E.PLAN_CACHE_KEEP


@pytest.mark.parametrize("flip", [False, True])
def test_several_input_buffers_equal_one(flip):
    """smap_plan_run_inputs: the frames of a launch in 2 / 4 separate buffers give the bits of one gathered buffer (the stem only
    indexes differently) -- what CoalescedPipeline relies on to run two callers' batches as one launch without a copy."""
    from exps.stage3_root2.config import cfg
    net = _small_net().to(DEV)
    net.precision = "x3"
    fp = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [cfg.DATASET.KEYPOINT.NUM + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
    eng = net.engine(4, 64, 96, torch.device(DEV), flip_pair=fp if flip else None)
    x = torch.randn(4, 3, 64, 96, generator=torch.Generator().manual_seed(9)).to(DEV)
    ref = eng.new_output()
    eng.run(x, out=ref)
    for n in (2, 4):
        parts = [x[i * (4 // n):(i + 1) * (4 // n)].clone() for i in range(n)]      # separate allocations
        out = eng.new_output()
        eng.run(parts, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), n
    with pytest.raises(ValueError):
        eng.run([x[:1], x[1:]])                                                     # unequal parts
    with pytest.raises(ValueError):
        eng.run([x[:1].clone(), x[1:2].clone(), x[2:3].clone()])                    # 3 does not divide 4


def test_arena_beyond_4_gib_matches_smaller_launches():
    """The reference's shipped setting (test.sh: --batch_size 16 --do_flip 1) as ONE 32-frame schedule: a 5.6 GiB arena whose
    later tensors lie beyond the first 4 GiB window.  Every launch addresses its input from the input's own window; the maps
    must equal what two 8-frame flip launches give to 2e-6 (heuristic tiles on both sides; the tile of a low-resolution layer
    still depends on M, so the sums may run in another order -- the split-precision rounding level, not bit-equality)."""
    import smap_amd.engine as E
    from exps.stage3_root2.config import cfg
    from smap_amd.model.smap import SMAP
    old = E._TILE_TABLE_X3
    E._TILE_TABLE_X3 = {}
    # (the two-input launches -- conv_cat, conv_relusum -- address both inputs from ONE window and are left out of schedules beyond 4 GiB: the
    #  16-frame side is built without them too, so that both sides run the same launches)
    saved = {k: os.environ.get(k) for k in ("SMAP_CAT", "SMAP_SKIPSUM")}
    os.environ.update(SMAP_CAT="0", SMAP_SKIPSUM="0")
    try:
        torch.manual_seed(0)
        net = SMAP(make_cfg((128, 208))).eval()
        net.load_state_dict(recipe_state_dict(net.state_dict()))
        net = net.to(DEV)
        net.precision = "x3"
        fp = list(cfg.DATASET.KEYPOINT.FLIP_ORDER) + [cfg.DATASET.KEYPOINT.NUM + c for c in cfg.DATASET.PAF.FLIP_CHANNEL]
        big = net.engine(16, 512, 832, torch.device(DEV), flip_pair=fp)
        assert big.graph.arena_bytes > (1 << 32)
        convs = [op for op in big.graph.ops if op.kind == E.OP_CONV]
        assert any(op.inp.off >= E.WINDOW for op in convs), "some launch must read from the second window"
        x = torch.randn(16, 3, 512, 832, generator=torch.Generator().manual_seed(21)).to(DEV)
        hb = [t.clone() for t in big.run(x)]
        assert big.status() == 0
        del big
        net.invalidate_engine()
        torch.cuda.empty_cache()
        small = net.engine(8, 512, 832, torch.device(DEV), flip_pair=fp)
        for j in range(2):
            hs = small.run(x[8 * j:8 * j + 8])
            for a, b, name in zip(hb, hs, ("hms", "det_d", "root_d")):
                a = a[8 * j:8 * j + 8]
                err = (a - b).abs().max().item() / b.abs().max().item()
                assert err <= 2e-6, (name, j, err)
    finally:
        E._TILE_TABLE_X3 = old
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
