#!/usr/bin/env python3
"""Per-kernel wave-state counters from a rocprofv3 --pmc pass (its own run, --kernel-trace only):
    python tools/prof_counters.py DIR/.../pmc_counter_collection.csv [kernel-name fragment ...]
Prints every counter summed over the matching dispatches and, where SQ_WAVE_CYCLES is present, as a share of it
(SQ_WAIT_ANY = parked in s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue-stalled, SQ_ACTIVE_INST_ANY = issuing)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frags = sys.argv[2:] or ["bottleneck"]
by = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    if any(f in name for f in frags):
        short = name.split("(")[0][-60:]
        by[short][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(short, r["Counter_Name"])] += 1
for k, c in by.items():
    calls = max(v for (kk, _), v in n.items() if kk == k)
    print(f"{k}  ({calls} dispatches)")
    wc = c.get("SQ_WAVE_CYCLES")
    for name, v in sorted(c.items()):
        print(f"    {name:28s} {v / calls:14.4g} per dispatch" + (f"   {v / wc:6.3f} of SQ_WAVE_CYCLES" if wc else ""))
