#!/bin/bash
# GPU visit 18: fused stem + max-pool kernel: parity, in-situ A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "stem_pool or small_schedule or full_size or module_forward" 2>&1 | tail -4
for i in 1 2; do
for v in "1" ""; do
  echo "-- SMAP_NO_STEMPOOL=$v"
  SMAP_NO_STEMPOOL=$v timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200 | tee -a $O/ab_stempool.log
done
done
