#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
echo "== full"; python tools/bench_conv.py --iters 30 2>/dev/null | tee $O/abl_full.log
for n in 1 2 8 9 10; do
  echo "== ablate $n (1 no loads, 2 no mfma, 8 no epilogue)"
  SMAP_HIP_LIB=$R/smap_amd/csrc/obj/libsmap_hip_abl$n.so python tools/bench_conv.py --iters 30 2>/dev/null | tee $O/abl_$n.log
done
