#!/bin/bash
# Round 2, GPU visit 7: shipping-config evidence (rocprofv3 stats + per-layer join, default bench lines of every config),
# then an in-situ coordinate descent over the x3 tile table.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python bench.py > $O/r2_07_bench_x3.json 2> $O/r2_07_bench_x3.err; tail -c 400 $O/r2_07_bench_x3.json
timeout 600 python bench.py --precision f16 > $O/r2_07_bench_f16.json 2>/dev/null; tail -c 300 $O/r2_07_bench_f16.json
for a in "--refine" "--flip" "--forward-only --batch 1" "--forward-only --batch 1 --graph" "--forward-only --batch 1 --precision f16" "--forward-only --batch 8"; do
  echo "-- bench.py $a" | tee -a $O/r2_07_configs.log
  timeout 300 python bench.py $a --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee -a $O/r2_07_configs.log
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_r2_07_x3_d1 -o smap -- python $R/bench.py --depth 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/rocprof_r2_07_d1.log 2>&1; echo "trace d1 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_r2_07_x3_d2 -o smap -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_r2_07_d2.log 2>&1; echo "trace d2 rc=$?"
cd $R
python tools/prof_export.py $O/prof_r2_07_x3_d2/smap_results.db $O/r2_07_x3_kernel_stats.csv
cp $O/prof_r2_07_x3_d2/smap_kernel_stats.csv $O/r2_07_x3_rocprofv3_stats_native.csv 2>/dev/null
SMAP_PRECISION=x3 python tools/prof_layers.py $O/prof_r2_07_x3_d1/smap_results.db 8 > $O/r2_07_x3_layers.txt 2>&1; tail -3 $O/r2_07_x3_layers.txt
rm -rf $O/prof_r2_07_x3_d1 $O/prof_r2_07_x3_d2
timeout 1200 python tools/insitu_tune.py --precision x3 --candidates tools/insitu_candidates_x3.json --out $O/tile_table_x3_insitu.json 2>&1 | tee $O/r2_07_insitu_x3.log | tail -40
