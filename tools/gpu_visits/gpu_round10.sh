#!/bin/bash
# GPU visit 10: weight-stationary persistent 1x1 kernel (conv1.hip): parity, micro-bench, in-situ A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "single_conv or halo or small_schedule" -p no:cacheprovider > $O/pytest_ws.log 2>&1; echo "rc=$?"; tail -12 $O/pytest_ws.log
echo "== microbench"
timeout 300 python tools/bench_conv.py --iters 30 --only L1,L3,L7,L4 2>&1 | grep -v amdgpu.ids | tee $O/mb_ws.log
timeout 300 python tools/bench_conv.py --iters 30 --only L1,L3,L7,L4 --tile-override L1:40,L3:40,L7:40,L4:40 2>&1 | grep -v amdgpu.ids | tee -a $O/mb_ws.log
timeout 300 python tools/bench_conv.py --iters 30 --only L1,L3,L7,L4 --tile-override L1:41,L3:41,L7:41,L4:41 2>&1 | grep -v amdgpu.ids | tee -a $O/mb_ws.log
echo "== in-situ A/B"
for v in "" "40" "41" "40 100000" "40 50000"; do
  set -- $v
  echo "-- SMAP_WS1=${1:-} MIN_M=${2:-}"
  SMAP_WS1=${1:-} SMAP_WS1_MIN_M=${2:-} timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260 | tee -a $O/ab_ws.log
done
