#!/bin/bash
# GPU visit 12: fold the weight-stationary 1x1 variants into the measured tile table, A/B in situ.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/autotune.py --halo 0.05 --iters 30 --out $O/tile_table_ws.json 2>&1 | grep -v amdgpu.ids | tee $O/autotune_ws.log
for i in 1 2; do
for tb in "" "$O/tile_table_ws.json"; do
  echo "-- table=${tb:-shipped}"
  SMAP_TILE_TABLE=$tb timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $O/ab_table_ws.log
done
done
