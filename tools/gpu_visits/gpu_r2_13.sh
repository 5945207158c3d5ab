#!/bin/bash
# Round 2, GPU visit 13: software-pipelined K loop (tiles 70..76): parity, cold single-layer timings vs the un-pipelined
# tiles, autotune incl. the new tiles, in-situ A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "single_conv" 2>&1 | tail -6 | tee $O/r2_13_tests.log
for t in 20 70 50 71 21 73 23 74; do
  echo "== x3 tile $t cold" | tee -a $O/r2_13_stream.log
  python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L2,L4,L7,L8 --tile-override L3:$t,L1:$t,L2:$t,L4:$t,L7:$t,L8:$t 2>/dev/null | tee -a $O/r2_13_stream.log
done
for t in 0 70 4 74 2 75; do
  echo "== f16 tile $t cold" | tee -a $O/r2_13_stream.log
  python tools/bench_conv.py --iters 30 --rotate 5 --only L3,L1,L2,L4,L7,L8 --tile-override L3:$t,L1:$t,L2:$t,L4:$t,L7:$t,L8:$t 2>/dev/null | tee -a $O/r2_13_stream.log
done
timeout 900 python tools/autotune.py --precision x3 --iters 20 --out $O/tile_table_x3_v4.json 2>&1 | tee $O/r2_13_autotune_x3.log | tail -2
for i in 1 2; do
  echo "-- x3 shipped table" | tee -a $O/r2_13_ab.log
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_13_ab.log
  echo "-- x3 v4 table (pipelined tiles as candidates)" | tee -a $O/r2_13_ab.log
  SMAP_TILE_TABLE_X3=$O/tile_table_x3_v4.json timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_13_ab.log
done
SMAP_TILE_TABLE_X3=$O/tile_table_x3_v4.json timeout 600 python -m pytest tests/test_e2e_parity_gpu.py -m gpu -q -p no:cacheprovider -k "split_precision and smooth" 2>&1 | tail -3 | tee -a $O/r2_13_tests.log
