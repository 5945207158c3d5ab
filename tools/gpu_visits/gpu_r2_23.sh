#!/bin/bash
# Round 2, GPU visit 23: the committed tree once more -- full GPU suite, smoke, default bench lines (x3, f16), rocprofv3 stats of both.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ) 2>&1 | tee $O/r2_23_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/r2_23_smoke.log
timeout 600 python bench.py > $O/r2_23_bench_x3.json 2>/dev/null; tail -c 200 $O/r2_23_bench_x3.json; echo
timeout 600 python bench.py --precision f16 > $O/r2_23_bench_f16.json 2>/dev/null; tail -c 200 $O/r2_23_bench_f16.json; echo
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_r2_23_x3 -o smap -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_r2_23_x3.log 2>&1; echo "trace x3 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_r2_23_f16 -o smap -- python $R/bench.py --precision f16 --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_r2_23_f16.log 2>&1; echo "trace f16 rc=$?"
cd $R
python tools/prof_export.py $O/prof_r2_23_x3/smap_results.db $O/r2_23_x3_kernel_stats.csv
python tools/prof_export.py $O/prof_r2_23_f16/smap_results.db $O/r2_23_f16_kernel_stats.csv
cp $O/prof_r2_23_x3/smap_kernel_stats.csv $O/r2_23_x3_rocprofv3_stats_native.csv 2>/dev/null
cp $O/prof_r2_23_f16/smap_kernel_stats.csv $O/r2_23_f16_rocprofv3_stats_native.csv 2>/dev/null
rm -rf $O/prof_r2_23_x3 $O/prof_r2_23_f16
