#!/bin/bash
# Round 3, GPU visit 9: the ops with a fused bilinear add (up4.out, up3.out): which tile, which epilogue pipeline depth?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m smap_amd.build > /dev/null 2>&1
python tools/build_ablate.py --epi-depth > /dev/null 2>&1
echo "== L20 = up4.out, L21 = up3.out (cold); epi1 = shipped epilogue (1 pass of loads ahead), epi2 = 2 passes ahead" | tee $O/v9_up_tiles.log
for lib in smap_amd/libsmap_hip.so smap_amd/csrc/obj/libsmap_hip_epi2.so; do
  for t in 20 50 51 24 0 52 54 53; do
    SMAP_HIP_LIB=$lib timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L20,L21 --tile-override L20:$t,L21:$t 2>&1 | grep -v amdgpu.ids | sed "s/^/$(basename $lib .so | sed s/libsmap_hip_*//): /" | tee -a $O/v9_up_tiles.log
  done
done
