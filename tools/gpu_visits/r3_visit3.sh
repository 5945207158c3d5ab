#!/bin/bash
# Round 3, GPU visit 3: weight tiles as contiguous pre-swizzled blocks (all three conv kernels): parity of every tile, then timings.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== backbone parity"
timeout 1200 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $O/v3_parity.log
echo "== bench, shipped table"
for i in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $O/v3_bench.log
done
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --depth 1 2>&1 | tail -1 | cut -c1-330 | tee -a $O/v3_bench.log
echo "== isolated cold timings: persistent tiles vs shipped"
timeout 900 python tools/autotune.py --precision x3 --convp 0.03 --iters 20 --out $O/tile_table_x3_convp.json 2>&1 | grep -v amdgpu.ids | tee $O/v3_autotune_convp.log
echo "== in situ with the persistent tiles"
for i in 1 2; do
  SMAP_TILE_TABLE_X3=$O/tile_table_x3_convp.json timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $O/v3_bench_convp.log
done
