#!/usr/bin/env python3
"""Full-size pin of the backbone oracle: the IMPORTED reference `model.smap.SMAP` (model/smap.py:313-421, inference branch
:403-419) run at BASELINE configs[1] -- 1 x 3 x 512 x 832, recipe weights (benchkit/recipe.py), input randn(seed 1234) --
in the authoring container; what travels is a digest of its three outputs:

    backbone_full.npz   per output (hms [43,128,208], det_d [14,..], root_d [1,..]):
                        *_chsum / *_chabs  per-channel sum and sum of |.| in float64
                        *_idx / *_val      1024 sampled flat positions (fixed RNG) and the fp32 values there
                        *_absmax           max |.|
                        x_sum, x_abssum    digest of the input (regenerated from the seed by the tests)

    python tests/golden/gen_golden_full.py

Tests: tests/test_oracle_cpu.py holds oracle/backbone_ref.py to <= 1e-5 of it (CPU, ~4 s), tests/test_backbone_gpu.py the HIP
split-precision path to <= 1e-4 (GPU).  Environment accommodations: as in gen_golden.py (easydict stand-in, nothing else).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import _install_reference, make_cfg  # noqa: E402
from recipe import recipe_state_dict  # noqa: E402

N_SAMPLES = 1024


def digest(name, t, rng):
    a = t[0].numpy()                                            # [C,h,w]
    flat = a.reshape(-1)
    idx = np.sort(rng.choice(flat.size, size=min(N_SAMPLES, flat.size), replace=False)).astype(np.int64)
    return {name + "_chsum": a.astype(np.float64).sum((1, 2)), name + "_chabs": np.abs(a.astype(np.float64)).sum((1, 2)),
            name + "_idx": idx, name + "_val": flat[idx].copy(), name + "_absmax": np.float64(np.abs(a).max()),
            name + "_shape": np.asarray(a.shape, np.int64)}


def main():
    _install_reference()
    from model.smap import SMAP
    torch.manual_seed(0)
    net = SMAP(make_cfg((128, 208))).eval()
    sd = recipe_state_dict(net.state_dict())
    net.load_state_dict(sd)
    x = torch.randn(1, 3, 512, 832, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        hms, det_d, root_d = net(x)
    rng = np.random.default_rng(20260927)
    out = {"x_sum": np.float64(x.double().sum()), "x_abssum": np.float64(x.double().abs().sum())}
    for name, t in (("hms", hms), ("det_d", det_d), ("root_d", root_d)):
        out.update(digest(name, t, rng))
    np.savez_compressed(os.path.join(HERE, "backbone_full.npz"), **out)
    print("backbone_full:", {k: (tuple(v.shape) if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
