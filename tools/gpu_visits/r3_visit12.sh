#!/bin/bash
# Round 3, GPU visit 12: persistent kernel with 64-half K tiles (128-byte activation rows = full cache lines) on the MFMA-heavy shapes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -p no:cacheprovider -k "x69" 2>&1 | tail -3 | tee $O/v12_parity.log
echo "== cold timings: 52 / 62 / 69 on layer3 3x3 (L2), layer3 c1 (L14), up2 skip1 (L13), layer3 c3 (L4), layer4 3x3 (L6)" | tee $O/v12_bk64.log
for t in 52 62 69; do
  timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L2,L14,L13,L4,L6 --tile-override L2:$t,L14:$t,L13:$t,L4:$t,L6:$t 2>&1 | grep -v amdgpu.ids | tee -a $O/v12_bk64.log
done
python tools/build_ablate.py --trace > /dev/null 2>&1
SMAP_HIP_LIB=smap_amd/csrc/obj/libsmap_hip_trace.so timeout 300 python tools/trace_convp.py L2:62 L2:69 L14:69 2>&1 | grep -v amdgpu.ids | tee $O/v12_trace.log
