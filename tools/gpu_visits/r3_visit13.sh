#!/bin/bash
# Round 3, GPU visit 13: is the persistent kernel's LDS-DMA issue rate a per-wave limit?  8 loader waves; loader priority.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/build_convp_variants.py 0 -DSMAP_CONVP_LOADER_PRIO=3 > /dev/null 2>&1
echo "== cold timings: tile 62 (4 loader waves) vs 70 (8 loader waves); then 62 with s_setprio 3 in the loaders" | tee $O/v13_loaders.log
for t in 62 70; do
  timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L2,L14,L13,L4,L3 --tile-override L2:$t,L14:$t,L13:$t,L4:$t,L3:$t 2>&1 | grep -v amdgpu.ids | tee -a $O/v13_loaders.log
done
for t in 62 70; do
SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_pabl0.so timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L2,L14,L13,L4,L3 --tile-override L2:$t,L14:$t,L13:$t,L4:$t,L3:$t 2>&1 | grep -v amdgpu.ids | sed 's/^/prio3: /' | tee -a $O/v13_loaders.log
done
