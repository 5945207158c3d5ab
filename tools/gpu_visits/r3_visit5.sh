#!/bin/bash
# Round 3, GPU visit 5: does the activation stream of a persistent workgroup scale with the bytes it keeps in flight?
# N = 64 layer (K 256 -> N 64 at 128x208), 128x64 tiles with 6 / 3 / 2 ring stages: activation DMA only, then everything.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/build_convp_variants.py 42 10 > /dev/null 2>&1
echo "== L19 (K256 -> N64 at 128x208): tiles 63 / 64 / 65 = 6 / 3 / 2 stages (80 / 32 / 16 KB of activations in flight), 25 = shipped" | tee $O/v5_inflight.log
for n in 42 10 0; do
  lib=smap_amd/csrc/obj/libsmap_hip_pabl$n.so
  [ $n = 0 ] && lib=smap_amd/libsmap_hip.so
  for t in 63 64 65 25; do
    [ $n != 0 ] && [ $t = 25 ] && continue
    SMAP_HIP_LIB=$lib timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L19 --tile-override L19:$t 2>&1 | grep -v amdgpu.ids | sed "s/^/abl $n: /" | tee -a $O/v5_inflight.log
  done
done
