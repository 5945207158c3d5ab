#!/bin/bash
# Round 2, GPU visit 15: tap-fastest K order for the 3x3 im2col path (L2 locality): parity + cold timings + bench A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "single_conv or small_schedule" 2>&1 | tail -4 | tee $O/r2_15_tests.log
for t in 52 0 20 50; do
  echo "== x3 tile $t cold (3x3 presets)" | tee -a $O/r2_15_3x3.log
  python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L2,L6,L8,L5 --tile-override L2:$t,L6:$t,L8:$t,L5:21 2>/dev/null | tee -a $O/r2_15_3x3.log
done
for t in 0 2 1; do
  echo "== f16 tile $t cold (3x3 presets)" | tee -a $O/r2_15_3x3.log
  python tools/bench_conv.py --iters 30 --rotate 5 --only L2,L6,L8 --tile-override L2:$t,L6:$t,L8:$t 2>/dev/null | tee -a $O/r2_15_3x3.log
done
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_15_ab.log
  timeout 300 python bench.py --no-cpu-baseline --precision f16 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_15_ab.log
done
timeout 900 python tools/autotune.py --precision x3 --iters 20 --out $O/tile_table_x3_v5.json 2>&1 | tee $O/r2_15_autotune_x3.log | tail -2
SMAP_TILE_TABLE_X3=$O/tile_table_x3_v5.json timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_15_ab.log
SMAP_TILE_TABLE_X3=$O/tile_table_x3_v5.json timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_15_ab.log
