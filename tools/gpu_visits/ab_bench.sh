#!/bin/bash
# In-situ A/B of environment settings on ONE box, interleaved over repetitions (boxes differ by a few per cent, so only numbers
# of one visit compare).  Every variant is one string of VAR=value assignments ("" = baseline):
#     bash tools/gpu_visits/ab_bench.sh gpurun_out/ab.log 2 "--no-cpu-baseline --steps 60" "" "SMAP_BLOCK=64:91" "SMAP_BLOCK=64:90 SMAP_LAUNCH_FRAMES=32"
# (replaces the ~60 one-off visit scripts of rounds 1-3; their results are in profiles/ and EXPERIMENTS.md)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
OUT=$1; REPS=$2; ARGS=$3; shift 3
mkdir -p "$(dirname "$OUT")"
for rep in $(seq 1 "$REPS"); do
  for v in "$@"; do
    env $v SMAP_BENCH_NO_LF0=1 timeout 400 python bench.py $ARGS 2>>"$OUT.err" | python -c "
import sys, json
tag = sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print(tag, round(d['value'], 1), 'fps', round(d['ms_per_step'], 3), 'ms/step', 'frames/launch', c.get('frames_per_launch'))
" "rep $rep [$v] $ARGS:" >> "$OUT"
  done
done
cat "$OUT"
