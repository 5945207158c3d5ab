#!/bin/bash
# Round 3, GPU visit 7: which layers over-fetch?  Per-op FETCH_SIZE / WRITE_SIZE of one x3 forward (separate PMC passes).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_v7 -o pmc -- python $R/bench.py --depth 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_v7.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_v7 -o pmc -- python $R/bench.py --depth 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_v7.log 2>&1; echo "write rc=$?"
cd $R
F=$(ls $O/pmc_fetch_v7/*counter_collection.csv $O/pmc_fetch_v7/*/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls $O/pmc_write_v7/*counter_collection.csv $O/pmc_write_v7/*/*counter_collection.csv 2>/dev/null | head -1)
python tools/prof_traffic_layers.py $F $W x3 > $O/v7_traffic_layers.txt 2>&1
tail -60 $O/v7_traffic_layers.txt
rm -rf $O/pmc_fetch_v7 $O/pmc_write_v7
