#!/bin/bash
# Round 3, GPU visit 10: new x3 table (own entries for the bilinear-add ops, persistent tiles where they won) in situ + full GPU tests.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $O/v10_bench.log
done
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $O/v10_gpu_tests.log
