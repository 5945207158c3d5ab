#!/bin/bash
# GPU visit 15: halo kernel with batched fragment reads: parity + micro-bench + in situ.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "single_conv or halo" -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/bench_conv.py --iters 30 --only L5,L8,L2,L6,L9,L10 --tile-override L5:36,L8:31,L2:36,L6:36,L9:36,L10:39 2>&1 | grep -v amdgpu.ids | tee $O/mb_halo2.log
timeout 300 python tools/bench_conv.py --iters 30 --only L5,L8,L2,L6,L9 --tile-override L5:50,L8:51,L2:57,L6:56,L9:56 2>&1 | grep -v amdgpu.ids | tee -a $O/mb_halo2.log
for i in 1 2; do
  timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $O/ab_halo2.log
done
