#!/bin/bash
# round 4, visit 13: in-situ A/B of the table with the staggered eight-wave halo tiles (44 / 45) against the previous table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v13; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_visits/ab_bench.sh $O/ab_halo8.log 3 "--no-cpu-baseline --steps 40 --warmup 6" "SMAP_TILE_TABLE_X3=tools/tile_table_x3_r3.json" ""
