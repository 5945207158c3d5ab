#!/bin/bash
# round 4, visit 4: the first-block variant of csrc/convb.hip (tile ids 92, 93): probe, tests, in-situ A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v4; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/debug/convb_probe.py 92 93 > $O/probe.log 2>&1; echo "probe rc $?" >> $O/probe.log
grep -v "^   " $O/probe.log | head -20
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "whole_bottleneck or fused_bottleneck_tail" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
bash tools/gpu_visits/ab_bench.sh $O/ab_block.log 2 "--no-cpu-baseline --steps 60" "" "SMAP_BLOCK=64:91" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:92" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93 SMAP_LAUNCH_FRAMES_X=1" > /dev/null
bash tools/gpu_visits/ab_bench.sh $O/ab_block_d1.log 1 "--no-cpu-baseline --steps 40 --depth 1 --launch-frames 0" "" "SMAP_BLOCK=64:91" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:92" > /dev/null
bash tools/gpu_visits/ab_bench.sh $O/ab_block_lf32.log 1 "--no-cpu-baseline --steps 64 --launch-frames 32" "" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93" > /dev/null
cat $O/ab_block.log $O/ab_block_d1.log $O/ab_block_lf32.log
