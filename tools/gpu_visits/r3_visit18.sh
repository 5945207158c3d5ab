#!/bin/bash
# Round 3, GPU visit 18: ResNet_top as one kernel in split precision (stem + max-pool fused): bit-exactness, then in situ A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -p no:cacheprovider -k "stem_pool" 2>&1 | tail -4 | tee $O/v18_parity.log
for i in 1 2; do
for sp in "" 1; do
  echo "-- SMAP_STEMPOOL=${sp:-0}" | tee -a $O/v18_ab_stempool.log
  SMAP_STEMPOOL=$sp timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $O/v18_ab_stempool.log
done
done
for sp in "" 1; do
  echo "-- depth 1 SMAP_STEMPOOL=${sp:-0}" | tee -a $O/v18_ab_stempool.log
  SMAP_STEMPOOL=$sp timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --depth 1 2>&1 | tail -1 | cut -c1-330 | tee -a $O/v18_ab_stempool.log
done
