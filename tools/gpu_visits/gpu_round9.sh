#!/bin/bash
# GPU visit 9: fold the halo-tiled 3x3 variants into the measured tile table, then A/B the two tables in situ.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "single_conv or halo" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/autotune.py --halo 0.04 --iters 30 --out $O/tile_table_halo.json 2>&1 | tee $O/autotune_halo.log
for i in 1 2; do
for tb in "" "$O/tile_table_halo.json"; do
  echo "-- table=${tb:-shipped}"
  SMAP_TILE_TABLE=$tb timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $O/ab_table.log
done
done
