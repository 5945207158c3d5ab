#!/bin/bash
# Round 2, GPU visit 2: more split-precision tile instances (parity), isolated-launch autotune of the x3 tiles,
# A/B of the tuned table inside the full bench, f16 regression check at the default step count.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "split_precision or rejects" 2>&1 | tail -5 | tee $O/r2_02_tests.log
timeout 900 python tools/autotune.py --precision x3 --iters 20 --out $O/tile_table_x3.json 2>&1 | tee $O/r2_02_autotune_x3.log | tail -45
for i in 1 2; do
  echo "-- x3 heuristic tiles" | tee -a $O/r2_02_bench.log
  timeout 300 python bench.py --precision x3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260 | tee -a $O/r2_02_bench.log
  echo "-- x3 tuned table" | tee -a $O/r2_02_bench.log
  SMAP_TILE_TABLE_X3=$O/tile_table_x3.json timeout 300 python bench.py --precision x3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260 | tee -a $O/r2_02_bench.log
done
echo "-- x3 tuned table depth 1" | tee -a $O/r2_02_bench.log
SMAP_TILE_TABLE_X3=$O/tile_table_x3.json timeout 300 python bench.py --precision x3 --depth 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260 | tee -a $O/r2_02_bench.log
echo "-- f16 default" | tee -a $O/r2_02_bench.log
timeout 300 python bench.py --precision f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260 | tee -a $O/r2_02_bench.log
timeout 300 python bench.py --precision f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260 | tee -a $O/r2_02_bench.log
