#!/bin/bash
# How much host does a rank need?  bench.py's REAL multi-rank branch with two ranks sharing cuda:0 (the hooks of
# tests/test_entry_gpu.py: SMAP_BENCH_SHARE_GPU, gloo collectives) under shrinking CPU affinity masks: 16 / 4 / 2 / 1 allowed
# CPUs for the PAIR of ranks.  The two ranks share one GPU, so the absolute frames/s are half a real rank's; what is read off is
# the GROWTH of the step time as the cores go away -- DESIGN.md section 7's ">= 2 host cores per rank" as a measurement.
#     bash tools/host_budget.sh [steps] > gpurun_out/host_budget.log
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
STEPS=${1:-30}
NCPU=$(python -c "import os; print(len(os.sched_getaffinity(0)))")
for n in $NCPU 4 2 1; do
  [ "$n" -gt "$NCPU" ] && continue
  last=$((n - 1))
  MASTER_ADDR=127.0.0.1 SMAP_BENCH_SHARE_GPU=1 SMAP_BENCH_BACKEND=gloo SMAP_BENCH_NO_LF0=1 timeout 600 taskset -c 0-$last \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29600 + n)) \
    bench.py --gpus 2 --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); h = d['config']['host_ms_per_step']
        print('cpus for 2 ranks: $n  ->', round(d['value'], 1), 'frames/s (2 ranks on ONE GPU)', round(d['ms_per_step'], 2), 'ms/step; per-rank process CPU ms/step', [round(x, 2) for x in h['process_cpu_per_rank']], 'enqueue', round(h['enqueue_and_records'], 2), 'wait', round(h['backpressure_wait'], 2), 'busiest', h['busiest_threads_cpu_ms'][:2])
"
done
