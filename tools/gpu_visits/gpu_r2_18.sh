#!/bin/bash
# Round 2, GPU visit 18: final validation of the tree as committed: full GPU suite, smoke, default bench lines.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tee $O/r2_18_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r2_18_smoke.log
( time timeout 600 python bench.py > $O/r2_18_bench_x3.json 2> $O/r2_18_bench_x3.err ) 2>&1 | tail -3; tail -c 300 $O/r2_18_bench_x3.json; echo
timeout 600 python bench.py --precision f16 > $O/r2_18_bench_f16.json 2>/dev/null; tail -c 200 $O/r2_18_bench_f16.json; echo
