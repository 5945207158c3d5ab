#!/bin/bash
# Round 2, GPU visit 24: BK = 16 staging for split precision (tiles 58, 59): parity, cold timings, autotune + in-situ A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "split_precision and single_conv" 2>&1 | tail -4 | tee $O/r2_24_tests.log
for t in 20 58 59; do
  echo "== x3 tile $t cold" | tee -a $O/r2_24_stream.log
  python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L2,L4,L7,L8 --tile-override L3:$t,L1:$t,L2:$t,L4:$t,L7:$t,L8:$t 2>/dev/null | tee -a $O/r2_24_stream.log
done
timeout 900 python tools/autotune.py --precision x3 --iters 20 --out $O/tile_table_x3_v7.json 2>&1 | tee $O/r2_24_autotune_x3.log | tail -2
for i in 1 2; do
  echo "-- x3 shipped table" | tee -a $O/r2_24_ab.log
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_24_ab.log
  echo "-- x3 v7 table (BK=16 candidates)" | tee -a $O/r2_24_ab.log
  SMAP_TILE_TABLE_X3=$O/tile_table_x3_v7.json timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_24_ab.log
done
python - <<'PY' | tee -a $O/r2_24_ab.log
import json
t = json.load(open('gpurun_out/tile_table_x3_v7.json'))
print('BK=16 picks:', {k: v for k, v in t.items() if int(v) in (58, 59)})
PY
