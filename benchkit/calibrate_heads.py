"""Generates benchkit/head_calibration.json (see benchkit/workload.py::people_state_dict): per key-point channel, the
bias that puts the TARGET-th highest strict 3x3 local maximum of frame 0 (seed 1234, 3x512x832) on the NMS threshold.
CPU only (fp32 oracle forward); run once: `python -m benchkit.calibrate_heads`."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
TARGET = 24


def main():
    from benchkit.workload import HEAD_KINDS, NJ, make_cfg, people_state_dict
    from smap_amd.model.smap import SMAP
    from oracle.backbone_ref import smap_forward
    torch.manual_seed(0)
    base = SMAP(make_cfg((128, 208))).state_dict()
    x = torch.randn(1, 3, 512, 832, generator=torch.Generator().manual_seed(1234))
    out = {}
    for kind in HEAD_KINDS:
        sd = people_state_dict(base, kind, calibration=[])
        with torch.no_grad():
            hms = smap_forward(sd, x)[0][0, :NJ]
        nb = F.unfold(hms[:, None], 3).view(NJ, 9, 126, 206)
        centre = nb[:, 4]
        others = torch.cat([nb[:, :4], nb[:, 5:]], 1).max(1).values
        bias = []
        for c in range(NJ):
            v = torch.sort(centre[c][centre[c] > others[c]], descending=True).values
            bias.append(float(0.2 * 255 - 0.5 * (v[TARGET - 1] + v[TARGET])))
        out[kind] = {"kpt_bias": bias, "target_peaks": TARGET}
    json.dump(out, open(os.path.join(ROOT, "benchkit", "head_calibration.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
