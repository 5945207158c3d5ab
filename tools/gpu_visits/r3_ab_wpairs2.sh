#!/bin/bash
# Round 3: pairs-vs-contiguous weight blocks chosen per launch size: threshold sweep on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -p no:cacheprovider -k "single_conv or schedule or golden" 2>&1 | tail -1
for rep in 1 2; do
for th in 0 256 512 1024 100000; do
  for a in "--forward-only --batch 1" ""; do
    v=$(SMAP_WPAIRS_MAX_TILES=$th timeout 300 python bench.py $a --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
    echo "max_tiles=$th | $a | $v"
  done
done
done
