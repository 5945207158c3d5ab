#!/bin/bash
# GPU visit 11: what bounds the streaming 1x1 kernel?  Ablations of conv1.hip on the K=256 / K=64 full-resolution layers.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -k "single_conv or halo or small_schedule" -p no:cacheprovider 2>&1 | tail -3
rm -f $O/abl_ws.log
for t in 40 41; do
echo "== full tile $t"; python tools/bench_conv.py --iters 30 --only L1,L3,L7,L4 --tile-override L1:$t,L3:$t,L7:$t,L4:$t 2>/dev/null | tee -a $O/abl_ws.log
for n in 1 2 4 8 6; do
  echo "== ablate $n tile $t (1 no loads, 2 no mfma, 4 no stores, 8 no epilogue)"
  SMAP_HIP_LIB=$R/smap_amd/csrc/obj/libsmap_hip_abl$n.so python tools/bench_conv.py --iters 30 --only L1,L3 --tile-override L1:$t,L3:$t 2>/dev/null | tee -a $O/abl_ws.log
done
done
