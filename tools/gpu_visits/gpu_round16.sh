#!/bin/bash
# GPU visit 16: ablations of the halo 3x3 kernel: what does a tap iteration cost?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T="L5:36,L8:31,L2:36,L6:36"
echo "== full"; python tools/bench_conv.py --iters 30 --only L5,L8,L2,L6 --tile-override $T 2>/dev/null | tee $O/abl_halo.log
for n in 1 2 8 16 17 19 27; do
  echo "== ablate $n (1 no DMA, 2 no MFMA, 8 no epilogue, 16 no ds_read)"
  SMAP_HIP_LIB=$R/smap_amd/csrc/obj/libsmap_hip_abl$n.so python tools/bench_conv.py --iters 30 --only L5,L8,L2,L6 --tile-override $T 2>/dev/null | tee -a $O/abl_halo.log
done
echo "== 8-wave 50/51/57/56"
T="L5:50,L8:51,L2:50,L6:56"
python tools/bench_conv.py --iters 30 --only L5,L8,L2,L6 --tile-override $T 2>/dev/null | tee -a $O/abl_halo.log
for n in 1 17 27; do
  echo "== ablate $n 8-wave"
  SMAP_HIP_LIB=$R/smap_amd/csrc/obj/libsmap_hip_abl$n.so python tools/bench_conv.py --iters 30 --only L5,L8,L2,L6 --tile-override $T 2>/dev/null | tee -a $O/abl_halo.log
done
