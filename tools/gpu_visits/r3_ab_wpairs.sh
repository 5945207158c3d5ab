#!/bin/bash
# Round 3: weight blocks of 32-half K tiles: one contiguous block per K tile (SMAP_WPAIRS=0) vs pairs in 128-byte rows (1), same box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
SMAP_WPAIRS=0 timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -p no:cacheprovider -k "single_conv" 2>&1 | tail -1
for rep in 1 2 3; do
for wp in 0 1; do
  for a in "--forward-only --batch 1" "--depth 1" ""; do
    v=$(SMAP_WPAIRS=$wp timeout 300 python bench.py $a --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
    echo "wpairs=$wp | $a | $v"
  done
done
done
