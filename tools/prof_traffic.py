#!/usr/bin/env python3
"""HBM traffic of one B = 8 forward from two rocprofv3 PMC passes (tools/gpu_visits/gpu_profile.sh with PMC=1):
    python tools/prof_traffic.py gpurun_out/pmc_fetch_<tag>/pmc_counter_collection.csv \
                                 gpurun_out/pmc_write_<tag>/pmc_counter_collection.csv profiles/<name>.json
Sums FETCH_SIZE / WRITE_SIZE (KB) over the conv launches (conv.hip / conv3.hip / conv1.hip kernels) of the LAST
complete forward in each trace.  FETCH_SIZE is doubled: gfx950 reports half of wide coalesced reads
(MI355X_MICROARCH.md; calibrated in round 1 on a 1x1 layer: 2 x FETCH_SIZE = input + residual + weights within 1 %)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchkit.buildhash import stamp  # noqa: E402

CONV = ("conv_igemm", "conv3x3_halo", "conv1x1_ws", "convp_kernel", "conv3_tail_kernel", "bottleneck_kernel", "bottleneck_first_kernel", "bottleneck128_kernel")


def total(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    stems = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"] or "stem_pool_kernel" in r["Kernel_Name"]]
    heads = [i for i, r in enumerate(rows) if "headsum" in r["Kernel_Name"] or "tapsum" in r["Kernel_Name"]]      # the three maps' final launches
    # the last forward that is complete: from its stem launch to its third headsum launch
    for s in reversed(stems):
        hs = [h for h in heads if h > s][:3]
        if len(hs) == 3:
            seg = rows[s:hs[-1] + 1]
            conv = [r for r in seg if any(k in r["Kernel_Name"] for k in CONV)]
            return sum(float(r["Counter_Value"]) for r in conv), len(conv)
    raise SystemExit("no complete forward in " + path)


def main():
    f_kb, n1 = total(sys.argv[1], "FETCH_SIZE")
    w_kb, n2 = total(sys.argv[2], "WRITE_SIZE")
    assert n1 == n2, (n1, n2)
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, with --kernel-trace only) over "
                     f"`python bench.py --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline`, summed over the {n1} conv launches "
                     f"(conv.hip + conv3.hip + convp.hip + convb.hip + convc.hip kernels) of one 8-frame forward",
           "fetch_size_kb_sum": f_kb, "write_size_kb_sum": w_kb,
           "hbm_read_bytes_per_batch": 2.0 * f_kb * 1024, "hbm_write_bytes_per_batch": w_kb * 1024,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; calibrated in "
                   "round 1 on a 1x1 layer: 2 x FETCH_SIZE = input + residual + weights bytes within 1 %)"}
    out.update(stamp())          # build_hash / commit: bench.py quotes the figures only for the tree they were measured on
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
