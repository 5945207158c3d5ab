#!/bin/bash
# Round 2, GPU visit 6: halo-tiled 3x3 kernel in split precision: parity, isolated autotune incl. halo + 8-wave tiles, in-situ A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "split_precision or rejects or halo" 2>&1 | tail -5 | tee $O/r2_06_tests.log
timeout 900 python tools/autotune.py --precision x3 --iters 20 --out $O/tile_table_x3_v3.json 2>&1 | tee $O/r2_06_autotune_x3.log | tail -3
for i in 1 2; do
  echo "-- x3 shipped table" | tee -a $O/r2_06_ab.log
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_06_ab.log
  echo "-- x3 v3 table (halo + 8-wave candidates)" | tee -a $O/r2_06_ab.log
  SMAP_TILE_TABLE_X3=$O/tile_table_x3_v3.json timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_06_ab.log
done
SMAP_TILE_TABLE_X3=$O/tile_table_x3_v3.json timeout 600 python -m pytest tests/test_e2e_parity_gpu.py -m gpu -q -p no:cacheprovider -k "split_precision" 2>&1 | tail -3 | tee -a $O/r2_06_tests.log
