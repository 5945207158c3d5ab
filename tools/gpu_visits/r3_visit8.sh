#!/bin/bash
# Round 3, GPU visit 8: what does the fused bilinear add cost in the conv epilogue (up4.out = 257 us vs 125 us without)?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/build_ablate.py 32 64 96 > /dev/null 2>&1
echo "== L3 = plain lateral, L20 = + bilinear add (tile 20, cold); ablate 32 no tap loads, 64 no index math, 96 both" | tee $O/v8_bilinear.log
for n in 0 32 64 96; do
  lib=smap_amd/csrc/obj/libsmap_hip_abl$n.so
  [ $n = 0 ] && lib=smap_amd/libsmap_hip.so
  SMAP_HIP_LIB=$lib timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L3,L20,L21 --tile-override L3:20,L20:20,L21:50 2>&1 | grep -v amdgpu.ids | sed "s/^/abl $n: /" | tee -a $O/v8_bilinear.log
done
