#!/usr/bin/env python3
"""In-situ tile search: isolated-launch timings do not transfer to the two-batches-in-flight pipeline,
so this tries alternative tiles for the most expensive layer shapes INSIDE the full bench and keeps what helps.

    python tools/insitu_tune.py [--steps 24] [--warmup 6] [--out gpurun_out/tile_table_insitu.json]
One bench.py run per trial (child process, SMAP_TILE_TABLE pointing at a trial table); coordinate descent over the
candidate list below; a change is kept when it beats the running best by more than --gain (default 0.4 %) twice."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANDIDATES = [      # shape key -> alternative conv.hip tiles to try
    ("8,128,208,256,256,1,1", [0, 1, 9]),
    ("8,128,208,64,256,1,1", [0, 1, 2]),
    ("8,32,52,256,1024,1,1", [0, 9, 2]),
    ("8,64,104,128,512,1,1", [4, 0, 1]),
    ("8,32,52,1024,256,1,1", [0, 9, 7]),
    ("8,128,208,256,64,1,1", [1, 7]),
    ("8,64,104,512,256,1,1", [0, 9]),
    ("8,128,208,256,768,1,1", [4, 5]),
    ("8,16,26,512,2048,1,1", [4, 0, 7]),
    ("8,64,104,512,128,1,1", [4, 1]),
]


PRECISION = "f16"


def bench(table_path, steps, warmup):
    env = dict(os.environ, **{"SMAP_TILE_TABLE_X3" if PRECISION == "x3" else "SMAP_TILE_TABLE": table_path})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup),
                        "--no-cpu-baseline", "--precision", PRECISION], capture_output=True, text=True, env=env, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1])["value"] if line else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--gain", type=float, default=0.004)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tile_table_insitu.json"))
    ap.add_argument("--candidates", default="", help="JSON file [[shape key, [tiles...]], ...] replacing the built-in list")
    ap.add_argument("--precision", choices=("f16", "x3"), default="f16", help="x3: search smap_amd/tile_table_x3.json")
    args = ap.parse_args()
    global CANDIDATES, PRECISION
    PRECISION = args.precision
    if args.candidates:
        CANDIDATES = [tuple(c) for c in json.load(open(args.candidates))]
    table = json.load(open(os.path.join(ROOT, "smap_amd", "tile_table_x3.json" if PRECISION == "x3" else "tile_table.json")))
    trial = args.out + ".trial"
    json.dump(table, open(trial, "w"))
    best = max(bench(trial, args.steps, args.warmup) for _ in range(2))
    print(f"baseline {best:.1f} fps", flush=True)
    for key, alts in CANDIDATES:
        if key not in table:
            print("skip (not in table)", key)
            continue
        for t in alts:
            if t == table[key]:
                continue
            cand = dict(table)
            cand[key] = t
            json.dump(cand, open(trial, "w"))
            v = bench(trial, args.steps, args.warmup)
            note = ""
            if v > best * (1 + args.gain):
                v2 = bench(trial, args.steps, args.warmup)          # confirm
                if v2 > best * (1 + args.gain):
                    table, best, note = cand, min(v, v2), "  <- kept"
                else:
                    note = f"  (not confirmed: {v2:.1f})"
            print(f"{key} tile {table.get(key) if note.endswith('kept') else t}: {v:.1f} fps{note}", flush=True)
    json.dump(table, open(args.out, "w"), indent=0, sort_keys=True)
    print(f"final {best:.1f} fps -> {args.out}")


if __name__ == "__main__":
    main()
