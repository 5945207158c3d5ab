#!/bin/bash
# Round 2, GPU visit 8: what bounds the streaming 1x1 layers in split precision?  Cold-cache (rotating arenas) single-layer
# timings of the 256->256 @128x208 lateral (L3) and the 64->256 c3 (L1) for every tile family incl. the new 4-stage tiles,
# then ablation builds (1 no loads, 2 no MFMA, 4 no stores, 8 no epilogue) on the shipped tile.  Also RefineNet e2e parity.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_e2e_parity_gpu.py tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "refinenet or (single_conv and (55 or 56 or 57))" 2>&1 | tail -4 | tee $O/r2_08_tests.log
for t in 20 50 54 55 56 57 24 21 25; do
  echo "== x3 tile $t warm / cold(rotate 3)" | tee -a $O/r2_08_stream.log
  python tools/bench_conv.py --x3 --iters 30 --only L3,L1 --tile-override L3:$t,L1:$t 2>/dev/null | tee -a $O/r2_08_stream.log
  python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1 --tile-override L3:$t,L1:$t 2>/dev/null | tee -a $O/r2_08_stream.log
done
for t in 4 0 50 55 56; do
  echo "== f16 tile $t warm / cold(rotate 5)" | tee -a $O/r2_08_stream.log
  python tools/bench_conv.py --iters 30 --only L3,L1 --tile-override L3:$t,L1:$t 2>/dev/null | tee -a $O/r2_08_stream.log
  python tools/bench_conv.py --iters 30 --rotate 5 --only L3,L1 --tile-override L3:$t,L1:$t 2>/dev/null | tee -a $O/r2_08_stream.log
done
for n in 1 2 4 8; do
  echo "== x3 tile 20 ablate $n (1 no loads, 2 no MFMA, 4 no stores, 8 no epilogue), cold" | tee -a $O/r2_08_stream.log
  SMAP_HIP_LIB=$R/smap_amd/csrc/obj/libsmap_hip_abl$n.so python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L2 --tile-override L3:20,L1:20,L2:20 2>/dev/null | tee -a $O/r2_08_stream.log
done
echo "== x3 tile 20 full, cold, L2 too" | tee -a $O/r2_08_stream.log
python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L2 --tile-override L3:20,L1:20,L2:20 2>/dev/null | tee -a $O/r2_08_stream.log
