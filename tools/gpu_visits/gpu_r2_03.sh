#!/bin/bash
# Round 2, GPU visit 3: full GPU suite, smoke, the new default bench line (x3 + CPU reference + e2e parity) and the f16 line,
# sub-batch experiment (B = 4 vs 8, heuristic tiles), PMC traffic of the x3 forward.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8 ) 2>&1 | tee $O/r2_03_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/r2_03_smoke.log
( time timeout 600 python bench.py > $O/r2_03_bench_x3.json 2> $O/r2_03_bench_x3.err ) 2>&1 | tail -3; tail -c 1500 $O/r2_03_bench_x3.json; tail -3 $O/r2_03_bench_x3.err
timeout 600 python bench.py --precision f16 > $O/r2_03_bench_f16.json 2> $O/r2_03_bench_f16.err; tail -c 600 $O/r2_03_bench_f16.json
for p in x3 f16; do
  for cfg in "8 2" "4 2" "4 3" "4 4" "2 4"; do
    set -- $cfg
    echo "-- $p no-table batch $1 depth $2" | tee -a $O/r2_03_subbatch.log
    SMAP_NO_TILE_TABLE=1 timeout 300 python bench.py --precision $p --batch $1 --depth $2 --steps $((192 / $1)) --warmup 6 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_03_subbatch.log
  done
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_x3 -o pmc -- python $R/bench.py --precision x3 --depth 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_x3.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_x3 -o pmc -- python $R/bench.py --precision x3 --depth 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_x3.log 2>&1; echo "write rc=$?"
cd $R
F=$(ls $O/pmc_fetch_x3/*counter_collection.csv $O/pmc_fetch_x3/*/*counter_collection.csv 2>/dev/null | head -1)
Wf=$(ls $O/pmc_write_x3/*counter_collection.csv $O/pmc_write_x3/*/*counter_collection.csv 2>/dev/null | head -1)
echo "F=$F W=$Wf"
[ -n "$F" ] && [ -n "$Wf" ] && python tools/prof_traffic.py $F $Wf $O/r2_03_hbm_traffic_x3.json | tail -12
rm -rf $O/pmc_fetch_x3 $O/pmc_write_x3
