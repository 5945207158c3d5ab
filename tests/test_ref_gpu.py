"""GPU: the reference's OWN extension sources (extensions/association.cpp + gpu/*.cu), built for
gfx950 by oracle/build_ref.py into oracle/_ref/, run beside the HIP path and the C oracle on the
same heat-maps.  This is what pins the oracle for nms / paf_score / group:
  * dapalib_ref_nofma (reference sources, -ffp-contract=off) == oracle == HIP, bit for bit;
  * dapalib_ref (default flags, FMA contraction as nvcc/hipcc do by default): identical peak
    sets / counts / limb assignments, float values within 1e-4 px.
A missing oracle/_ref/*.so FAILS these tests (it used to skip): without them the association oracle is unpinned, and
that must turn the suite red, not quiet.  Build them with `python oracle/build_ref.py` where /root/reference exists
(__graft_entry__.build() does); the .so files travel with the repo snapshot to the GPU box.
Scenes: synthetic 0..20-person scenes, pure noise (127-peak cap) AND maps made by the real network."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from helpers import synth_scene, noise_scene
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
DEV = "cuda:0"


def _load(name):
    if not os.path.exists(os.path.join(REF_DIR, name + ".so")):
        pytest.fail(f"oracle/_ref/{name}.so is missing: the association oracle would be unpinned. Run "
                    f"`python oracle/build_ref.py` in a container that has /root/reference (it travels with the snapshot).")
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    return importlib.import_module(name)


_NET = []


def _network_scenes():
    """Two frames of REAL network output (HIP backbone, calibrated recipe heads: ~24 peaks per key-point channel,
    PAF channels as the network makes them), scaled as test.py:111-112 does."""
    if not _NET:
        from helpers import make_cfg
        from benchkit.workload import people_state_dict
        from smap_amd.model.smap import SMAP
        import dapalib
        for kind, seed in (("smooth", 1234), ("noise", 77)):
            torch.manual_seed(0)
            net = SMAP(make_cfg((128, 208))).eval()
            net.load_state_dict(people_state_dict(net.state_dict(), kind))
            x = torch.randn(1, 3, 512, 832, generator=torch.Generator().manual_seed(seed))
            hms, _, rd = net.to(DEV)(x.to(DEV))
            dapalib.scale_hms_(hms)
            _NET.append((hms[0].cpu().numpy().copy(), rd[0, 0].cpu().numpy().copy()))
            del net
    return list(_NET)


def _scenes():
    sc = [synth_scene(k, seed=300 + k)[:2] for k in (0, 1, 3, 8, 20)]
    sc += [synth_scene(6, seed=9, noise=0.05, drop=0.3)[:2], noise_scene(5), noise_scene(6, amp=0.4)]
    return sc + _network_scenes()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _ref_connect(mod, hms, rd):
    out = mod.connect(torch.from_numpy(hms).to(DEV), torch.from_numpy(rd), 2, True)
    return out.numpy() if out.dim() == 3 else np.zeros((0, 15, 4), np.float32)


def test_reference_nofma_build_equals_oracle_and_hip_bitwise():
    import dapalib
    ref = _load("dapalib_ref_nofma")
    for i, (hms, rd) in enumerate(_scenes()):
        cands, pafs = ref.extract(torch.from_numpy(hms).to(DEV))
        opk = O.nms(hms)
        osc = O.paf_score(hms, opk)
        for j in range(15):
            n = int(opk[j, 0, 0])
            assert tuple(cands[j].shape) == (n, 3), (i, j)
            assert np.array_equal(bits(cands[j].numpy()), bits(opk[j, 1:1 + n])), (i, j)
        pairs = [0, 1, 0, 2, 0, 9, 9, 10, 10, 11, 0, 3, 3, 4, 4, 5, 2, 12, 12, 13, 13, 14, 2, 6, 6, 7, 7, 8]
        for l in range(14):
            na, nb = int(opk[pairs[2 * l], 0, 0]), int(opk[pairs[2 * l + 1], 0, 0])
            assert np.array_equal(bits(pafs[l].numpy()), bits(osc[l, :na, :nb])), (i, l)
        want = _ref_connect(ref, hms, rd)
        ob = O.group(opk, osc, rd)
        assert want.shape == ob.shape and np.array_equal(bits(want), bits(ob)), i
        got = dapalib.connect(torch.from_numpy(hms).to(DEV), torch.from_numpy(rd), 2, True)
        got = got.numpy() if got.dim() == 3 else np.zeros((0, 15, 4), np.float32)
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), i


def test_reference_default_build_same_decisions():
    import dapalib
    ref = _load("dapalib_ref")
    worst = 0.0
    for i, (hms, rd) in enumerate(_scenes()):
        want = _ref_connect(ref, hms, rd)
        got = dapalib.connect(torch.from_numpy(hms).to(DEV), torch.from_numpy(rd), 2, True)
        got = got.numpy() if got.dim() == 3 else np.zeros((0, 15, 4), np.float32)
        assert got.shape == want.shape, i
        assert np.array_equal(got[:, :, 3] > 0, want[:, :, 3] > 0), i           # same joints found
        assert np.array_equal(got[:, :, 3], want[:, :, 3]), i                   # same peak (its score is a copy)
        assert np.array_equal(got[:, :, :2].astype(np.int32), want[:, :, :2].astype(np.int32)), i
        if got.size:
            worst = max(worst, float(np.abs(got - want).max()))
    assert worst < 1e-4, worst
