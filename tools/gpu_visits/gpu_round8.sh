#!/bin/bash
# GPU visit 8: halo-tiled 3x3 kernel (conv3.hip): parity cases, per-layer micro-bench vs the shipped tiles, in-situ A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest conv cases"; timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "single_conv or halo" -p no:cacheprovider > $O/pytest_halo.log 2>&1; echo "rc=$?"; tail -8 $O/pytest_halo.log
echo "== microbench shipped tiles"; timeout 300 python tools/bench_conv.py --iters 30 --only L5,L8,L2,L6,L9,L10 --tile-override L5:2,L8:0,L2:9,L6:7,L9:2,L10:8 2>&1 | tee $O/mb_halo.log
for set in "L5:30,L8:30,L2:30,L6:30,L9:30,L10:30" "L5:32,L8:32,L2:32,L6:32,L9:32,L10:32" "L5:34,L8:34,L2:34,L6:34,L9:34,L10:34" "L5:36,L8:36,L2:36,L6:36,L9:36,L10:36" "L8:31,L2:31,L6:31" "L8:33,L2:33,L6:33" "L8:35,L2:35,L6:35" "L8:37,L2:37,L6:37"; do
  only=$(echo $set | sed 's/:[0-9]*//g')
  timeout 300 python tools/bench_conv.py --iters 30 --only $only --tile-override $set 2>&1 | tee -a $O/mb_halo.log
done
echo "== in-situ A/B"
for h in "" "16" "32" "16 1" "32 1" "16:64 1"; do
  set -- $h
  echo "-- SMAP_HALO3=${1:-} DEEP=${2:-}"
  SMAP_HALO3=${1:-} SMAP_HALO3_DEEP=${2:-} timeout 300 python bench.py --steps 36 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | tee -a $O/ab_halo.log
done
