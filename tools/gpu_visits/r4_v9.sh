#!/bin/bash
# round 4, visit 9: cheap in-situ A/Bs on the new default schedule: fused stem + max-pool, three backbones in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v9; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_visits/ab_bench.sh $O/ab_misc.log 2 "--no-cpu-baseline --steps 60" "" "SMAP_STEMPOOL=1" > /dev/null
bash tools/gpu_visits/ab_bench.sh $O/ab_depth3.log 2 "--no-cpu-baseline --steps 60 --depth 3" "" > /dev/null
cat $O/ab_misc.log $O/ab_depth3.log
