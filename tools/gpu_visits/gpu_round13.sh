#!/bin/bash
# GPU visit 13: 8-wave halo variants (tile ids 50..57): parity, autotune of the 3x3 shapes, in-situ A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "single_conv or halo" -p no:cacheprovider 2>&1 | tail -3
SMAP_AUTOTUNE_ONLY3=1 timeout 600 python tools/autotune.py --halo 0.03 --iters 30 --out $O/tile_table_8w.json 2>&1 | grep -v amdgpu.ids | tee $O/autotune_8w.log
for i in 1 2; do
for tb in "" "$O/tile_table_8w.json"; do
  echo "-- table=${tb:-shipped}"
  SMAP_TILE_TABLE=$tb timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $O/ab_table_8w.log
done
done
