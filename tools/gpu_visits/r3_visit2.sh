#!/bin/bash
# Round 3, GPU visit 2: where does a persistent conv workgroup spend its life?  Phase stamps + ablated builds.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/build_ablate.py --trace > /dev/null 2>&1
python tools/build_convp_variants.py 1 2 4 8 3 > /dev/null 2>&1
echo "== phase stamps" | tee $O/v2_trace.log
SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_trace.so timeout 300 python tools/trace_convp.py L3:60 L13:60 L14:62 L2:62 L11:60 L1:60 2>&1 | grep -v amdgpu.ids | tee -a $O/v2_trace.log
echo "== ablations (cold, 3 arenas)" | tee $O/v2_ablate.log
for n in 0 1 2 4 8 3; do
  lib=smap_amd/csrc/obj/libsmap_hip_pabl$n.so
  [ $n = 0 ] && lib=smap_amd/libsmap_hip.so
  echo "-- ablate $n" | tee -a $O/v2_ablate.log
  SMAP_HIP_LIB=$lib timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L3,L13,L14,L2,L11,L1 --tile-override L3:60,L13:60,L14:62,L2:62,L11:60,L1:60 2>&1 | grep -v amdgpu.ids | tee -a $O/v2_ablate.log
done
