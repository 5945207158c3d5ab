#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, short bench, rocprofv3 kernel stats.
# Everything is logged under gpurun_out/ (merged back by gpurun).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $O/rocminfo.txt
nproc > $O/nproc.txt
echo "== smoke" ; timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -5 $O/smoke.log
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -60 $O/pytest_gpu.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
echo "== bench" ; timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > $O/bench.log 2>&1 ; echo "bench rc=$?" ; tail -3 $O/bench.log
echo "== rocprof" ; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o smap -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof.log 2>&1 ; echo "rocprof rc=$?" ; tail -3 $O/rocprof.log
find $O/prof -name "*stats*" | head ; for f in $(find $O/prof -name "*kernel_stats.csv" | head -1); do head -25 $f; done
fi
