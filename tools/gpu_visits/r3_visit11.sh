#!/bin/bash
# Round 3, GPU visit 11: full cold re-tune of the x3 table now that weight tiles are contiguous blocks (halo and persistent tiles included).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python tools/autotune.py --precision x3 --cold --iters 15 --out $O/tile_table_x3_retuned.json 2>&1 | grep -v amdgpu.ids | tee $O/v11_autotune_x3_cold.log | tail -45
for i in 1 2; do
for tb in "" "$O/tile_table_x3_retuned.json"; do
  echo "-- table=${tb:-shipped}" | tee -a $O/v11_ab_table.log
  SMAP_TILE_TABLE_X3=$tb timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $O/v11_ab_table.log
done
done
