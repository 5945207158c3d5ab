#!/usr/bin/env python3
"""MFMA pipe utilisation of one forward from a rocprofv3 PMC pass (its own run, with --kernel-trace only):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o pmc -- \
        python bench.py --depth 1 --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline
    python tools/prof_mfma.py DIR/.../pmc_counter_collection.csv [profiles/mfma_utilisation_x3.json [FRAMES_PER_LAUNCH]]
(FRAMES_PER_LAUNCH = 16 for a pass over `bench.py --depth 1 --steps 2 --warmup 2`, the launches the default pipeline issues.)
Over the conv launches of the LAST complete forward: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8
XCDs x 1024 SIMDs).  depth 1 so that kernels do not overlap."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchkit.buildhash import stamp  # noqa: E402

CONV = ("conv_igemm", "conv3x3_halo", "convp_kernel", "conv3_tail_kernel", "bottleneck_kernel", "bottleneck_first_kernel", "bottleneck128_kernel")
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    by[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
ids = sorted(by)
stems = [i for i in ids if "stem_kernel" in by[i]["name"]]
for last in reversed(stems):
    seg = [i for i in ids if i >= last]
    heads = [i for i in seg if "headsum" in by[i]["name"] or "tapsum" in by[i]["name"]][:3]
    if len(heads) == 3:
        break
seg = [i for i in seg if i <= heads[-1]]
conv = [i for i in seg if any(k in by[i]["name"] for k in CONV)]
mf = sum(by[i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for i in conv)
ga = sum(by[i].get("GRBM_GUI_ACTIVE", 0) for i in conv)
print(f"x3 depth 1: launches {len(seg)} conv {len(conv)} MFMA busy cycles (sum over SIMDs) {mf:.4g} GRBM_GUI_ACTIVE over conv kernels {ga:.4g}")
print(f"   (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs: {ga:.4g} / 8 = {ga / 8:.4g} cycles = {ga / 8 / 2.4e6:.2f} ms at 2.4 GHz, the serial conv time of one forward)")
print(f"   MFMA pipe utilisation over the conv kernels = busy / (GUI_ACTIVE / 8 x 1024 SIMDs) = {mf / (ga / 8 * 1024):.3f}")
blk = [i for i in conv if "bottleneck" in by[i]["name"]]
if blk:
    mfb = sum(by[i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for i in blk)
    gab = sum(by[i].get("GRBM_GUI_ACTIVE", 0) for i in blk)
    print(f"   whole-Bottleneck launches alone ({len(blk)}): {mfb / (gab / 8 * 1024):.3f}")
if len(sys.argv) > 2:          # the figure bench.py quotes next to its arithmetic pipe_frac (roofline.pipe_frac_counters)
    import json
    fpl = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cmd = "--depth 1 --launch-frames 0 --steps 2 --warmup 1" if fpl == 8 else "--depth 1 --steps 2 --warmup 2"
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE (own pass, --kernel-trace only) over `python bench.py "
                         f"{cmd} --no-cpu-baseline`: MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) over the "
                         f"{len(conv)} conv launches of one {fpl}-frame forward (tools/prof_mfma.py)",
               "pipe_utilisation": mf / (ga / 8 * 1024), "conv_launches": len(conv), "frames_per_launch": fpl, "serial_conv_ms_at_2.4GHz": ga / 8 / 2.4e6,
               "whole_block_launches_pipe_utilisation": (mfb / (gab / 8 * 1024)) if blk else None, **stamp()}, open(sys.argv[2], "w"), indent=1)
