#!/bin/bash
# Round 3, GPU visit 4: decompose the persistent kernel's time: DMA only (A / W / both), epilogue only (with / without stores).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $O/v4_parity.log
python tools/build_convp_variants.py 10 26 42 3 7 2 1 > /dev/null 2>&1
echo "== ablations (cold, 3 arenas): 10 DMA only, 26 W DMA only, 42 A DMA only, 3 epilogue only, 7 epilogue without stores, 2 no MFMA, 1 no DMA" | tee $O/v4_ablate.log
for n in 0 10 26 42 3 7 2 1; do
  lib=smap_amd/csrc/obj/libsmap_hip_pabl$n.so
  [ $n = 0 ] && lib=smap_amd/libsmap_hip.so
  echo "-- ablate $n" | tee -a $O/v4_ablate.log
  SMAP_HIP_LIB=$lib timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L3,L14,L11,L1 --tile-override L3:60,L14:62,L11:60,L1:60 2>&1 | grep -v amdgpu.ids | tee -a $O/v4_ablate.log
done
echo "== same, tile 61 / 62 on the lateral" | tee -a $O/v4_ablate.log
for n in 0 10 42 3; do
  lib=smap_amd/csrc/obj/libsmap_hip_pabl$n.so
  [ $n = 0 ] && lib=smap_amd/libsmap_hip.so
  for t in 61 62; do
  SMAP_HIP_LIB=$lib timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L3 --tile-override L3:$t 2>&1 | grep -v amdgpu.ids | sed "s/^/abl $n: /" | tee -a $O/v4_ablate.log
  done
done
