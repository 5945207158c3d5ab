#!/bin/bash
# Round 2, GPU visit 14: pipelined K loop WITH the LDS-DMA pieces interleaved among the MFMAs (tiles 70..76)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "single_conv" 2>&1 | tail -4 | tee $O/r2_14_tests.log
for t in 20 70 50 71 52 72 21 73 23 74; do
  echo "== x3 tile $t cold" | tee -a $O/r2_14_stream.log
  python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L2,L4,L7,L8 --tile-override L3:$t,L1:$t,L2:$t,L4:$t,L7:$t,L8:$t 2>/dev/null | tee -a $O/r2_14_stream.log
done
for t in 0 70 4 74 2 75; do
  echo "== f16 tile $t cold" | tee -a $O/r2_14_stream.log
  python tools/bench_conv.py --iters 30 --rotate 5 --only L3,L1,L2,L4,L7,L8 --tile-override L3:$t,L1:$t,L2:$t,L4:$t,L7:$t,L8:$t 2>/dev/null | tee -a $O/r2_14_stream.log
done
