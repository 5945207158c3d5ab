#!/bin/bash
# rocprofv3 kernel trace of bench.py (+ optional HBM traffic PMC passes) -> gpurun_out/prof_<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-cur}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_$TAG -o smap -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_$TAG.log 2>&1; echo "trace rc=$?"
if [ "${PMC:-0}" = "1" ]; then
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_$TAG.log 2>&1; echo "fetch rc=$?"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_$TAG.log 2>&1; echo "write rc=$?"
fi
