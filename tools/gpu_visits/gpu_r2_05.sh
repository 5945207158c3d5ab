#!/bin/bash
# Round 2, GPU visit 5: eight-wave workgroup variants (tile ids 50..54): parity, isolated autotune (x3 and f16), in-situ A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "single_conv" 2>&1 | tail -5 | tee $O/r2_05_tests.log
timeout 900 python tools/autotune.py --precision x3 --iters 20 --out $O/tile_table_x3_v2.json 2>&1 | tee $O/r2_05_autotune_x3.log | tail -3
timeout 900 python tools/autotune.py --precision f16 --iters 20 --out $O/tile_table_f16_v2.json 2>&1 | tee $O/r2_05_autotune_f16.log | tail -3
for i in 1 2; do
  echo "-- x3 shipped table" | tee -a $O/r2_05_ab.log
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_05_ab.log
  echo "-- x3 v2 table (8-wave candidates)" | tee -a $O/r2_05_ab.log
  SMAP_TILE_TABLE_X3=$O/tile_table_x3_v2.json timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_05_ab.log
  echo "-- f16 shipped table" | tee -a $O/r2_05_ab.log
  timeout 300 python bench.py --precision f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_05_ab.log
  echo "-- f16 v2 table (isolated autotune incl. 8-wave)" | tee -a $O/r2_05_ab.log
  SMAP_TILE_TABLE=$O/tile_table_f16_v2.json timeout 300 python bench.py --precision f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_05_ab.log
done
