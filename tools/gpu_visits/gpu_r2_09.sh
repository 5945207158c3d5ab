#!/bin/bash
# Round 2, GPU visit 9: phase-stagger experiment (co-resident workgroups start half a tile time apart)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for s in "0" "300,8,512" "600,8,512" "1000,8,512" "600,0,512" "600,3,512" "600,8,100000" "600,0,100000"; do
  echo "== x3 tile 20 cold, SMAP_STAGGER=$s" | tee -a $O/r2_09_stagger.log
  SMAP_STAGGER=$s python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L7 --tile-override L3:20,L1:20,L7:20 2>/dev/null | tee -a $O/r2_09_stagger.log
done
for s in "0" "300,8,512" "600,8,512" "600,0,512"; do
  echo "== f16 table tiles cold, SMAP_STAGGER=$s" | tee -a $O/r2_09_stagger.log
  SMAP_STAGGER=$s python tools/bench_conv.py --iters 30 --rotate 5 --only L3,L1,L7 --tile-override L3:4,L1:4,L7:4 2>/dev/null | tee -a $O/r2_09_stagger.log
done
for s in "0" "300,8,512" "600,8,512" "600,0,512"; do
  echo "== full bench x3, SMAP_STAGGER=$s" | tee -a $O/r2_09_stagger.log
  SMAP_STAGGER=$s timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_09_stagger.log
  SMAP_STAGGER=$s timeout 300 python bench.py --no-cpu-baseline --depth 1 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_09_stagger.log
done
