#!/bin/bash
# Round 2, GPU visit 27: which runtime setting parks the HIP runtime's busy helper thread? (host CPU per step at N = 1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for e in "X=0" "AMD_DIRECT_DISPATCH=0" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=0" "GPU_MAX_HW_QUEUES=2" "HIP_LAUNCH_BLOCKING=0 ROC_CPU_WAIT_FOR_SIGNAL=0" "HSA_ENABLE_SDMA=0"; do
  echo "-- $e" | tee -a $O/r2_27_host_env.log
  env $e timeout 300 python bench.py --no-cpu-baseline --steps 16 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   fps %.1f' % d['value'], d['config']['host_ms_per_step'])" | tee -a $O/r2_27_host_env.log
done
