#!/bin/bash
# Round 3, GPU visit 6: do activation (HBM) and weight (L2) LDS-DMA streams overlap when different waves issue them?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -p no:cacheprovider -k "x66 or x68 or x63" 2>&1 | tail -3 | tee $O/v6_parity.log
python tools/build_convp_variants.py 10 2 > /dev/null 2>&1
echo "== lateral (L3) and up3 skip1 (L11): tile 60 = unified loaders, 66 = 2 activation + 2 weight waves; 62 / 68 the same for 128x128" | tee $O/v6_split.log
for n in 10 2 0; do
  lib=smap_amd/csrc/obj/libsmap_hip_pabl$n.so
  [ $n = 0 ] && lib=smap_amd/libsmap_hip.so
  for t in 60 66 62 68; do
    SMAP_HIP_LIB=$lib timeout 300 python tools/bench_conv.py --x3 --rotate 3 --iters 30 --only L3,L11,L13 --tile-override L3:$t,L11:$t,L13:$t 2>&1 | grep -v amdgpu.ids | sed "s/^/abl $n: /" | tee -a $O/v6_split.log
  done
done
