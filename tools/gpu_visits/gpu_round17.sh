#!/bin/bash
# GPU visit 17: paired 1x1 launches (u_skip|skip1, skip2|cross_conv, downsample|c1): parity + in-situ A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
for i in 1 2; do
for v in "1" ""; do
  echo "-- SMAP_NO_PAIRS=$v"
  SMAP_NO_PAIRS=$v timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $O/ab_pairs.log
done
done
