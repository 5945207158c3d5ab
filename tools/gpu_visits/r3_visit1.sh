#!/bin/bash
# Round 3, GPU visit 1: parity of the persistent wave-specialised conv kernel (convp.hip, tiles 60..62), cold isolated
# timings against the shipped tiles, and the in-situ A/B of the table that takes the winners.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke case"
timeout 180 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -k "x3-2x16x24x64x256x1x1x60" -p no:cacheprovider 2>&1 | tail -5 | tee $O/v1_smoke.log
if ! grep -q "1 passed" $O/v1_smoke.log; then
  echo "smoke case failed: running the remaining convp cases for the failure pattern, then stopping"
  timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "x60 or x61 or x62" -p no:cacheprovider 2>&1 | tail -40 | tee $O/v1_parity.log
  exit 1
fi
echo "== parity"
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "x60 or x61 or x62" -p no:cacheprovider 2>&1 | tail -15 | tee $O/v1_parity.log
echo "== isolated cold timings"
timeout 900 python tools/autotune.py --precision x3 --convp 0.03 --iters 20 --out $O/tile_table_x3_convp.json 2>&1 | tee $O/v1_autotune_convp.log
echo "== in situ"
for i in 1 2; do
for tb in "" "$O/tile_table_x3_convp.json"; do
  echo "-- table=${tb:-shipped}" | tee -a $O/v1_ab_table.log
  SMAP_TILE_TABLE_X3=$tb timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee -a $O/v1_ab_table.log
done
done
for tb in "" "$O/tile_table_x3_convp.json"; do
  echo "-- depth 1 table=${tb:-shipped}" | tee -a $O/v1_ab_table.log
  SMAP_TILE_TABLE_X3=$tb timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --depth 1 2>&1 | tail -1 | cut -c1-400 | tee -a $O/v1_ab_table.log
done
