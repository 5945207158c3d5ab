#!/bin/bash
# Round 2, GPU visit 19: direct-A 1x1 kernel (tiles 80, 81): parity, cold single-layer timings, autotune + in-situ A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "single_conv or rejects" 2>&1 | tail -6 | tee $O/r2_19_tests.log
for t in 20 80 81; do
  echo "== x3 tile $t cold" | tee -a $O/r2_19_stream.log
  python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L4,L7 --tile-override L3:$t,L1:$t,L4:$t,L7:$t 2>/dev/null | tee -a $O/r2_19_stream.log
done
for t in 4 0 80 81; do
  echo "== f16 tile $t cold" | tee -a $O/r2_19_stream.log
  python tools/bench_conv.py --iters 30 --rotate 5 --only L3,L1,L4,L7 --tile-override L3:$t,L1:$t,L4:$t,L7:$t 2>/dev/null | tee -a $O/r2_19_stream.log
done
timeout 900 python tools/autotune.py --precision x3 --iters 20 --out $O/tile_table_x3_v6.json 2>&1 | tee $O/r2_19_autotune_x3.log | tail -2
for i in 1 2; do
  echo "-- x3 shipped table" | tee -a $O/r2_19_ab.log
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_19_ab.log
  echo "-- x3 v6 table (direct-A candidates)" | tee -a $O/r2_19_ab.log
  SMAP_TILE_TABLE_X3=$O/tile_table_x3_v6.json timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_19_ab.log
done
python - <<'PY' | tee -a $O/r2_19_ab.log
import json
t = json.load(open('gpurun_out/tile_table_x3_v6.json'))
print('direct-A picks:', {k: v for k, v in t.items() if int(v) in (80, 81)})
PY
