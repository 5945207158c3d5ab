#!/bin/bash
# Round 3, final GPU visit: the state that is committed -- tests, default bench line (with cpu_baseline + e2e parity), the other
# configurations, rocprofv3 kernel stats + per-layer join, PMC traffic (depth 2, as the bench runs) and MFMA pipe counters.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
timeout 300 python bench.py 2>&1 | tail -1 > $O/r3_final_bench_x3.json; cut -c1-300 $O/r3_final_bench_x3.json
timeout 300 python bench.py --precision f16 --no-cpu-baseline 2>&1 | tail -1 > $O/r3_final_bench_f16.json; cut -c1-200 $O/r3_final_bench_f16.json
{
for a in "--launch-frames 0" "--flip" "--refine" "--forward-only --batch 1" "--depth 1" "--depth 1 --launch-frames 0"; do
  echo "-- bench.py $a --no-cpu-baseline"
  timeout 300 python bench.py $a --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
done
echo "-- bench.py --flip (with the flip parity block)"
timeout 400 python bench.py --flip 2>&1 | tail -1 > $O/r3_final_bench_x3_flip.json; python -c "import json; d=json.load(open('gpurun_out/r3_final_bench_x3_flip.json')); print(d['value'], json.dumps(d['config'].get('e2e_parity')))"
} 2>&1 | tee $O/r3_final_bench_other_configs.log
cd /tmp && export TMPDIR=/tmp
# (a) the default command (two 8-frame steps per 16-frame launch, two launches in flight): rocprofv3's own --stats table
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3_default -o smap -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/rocprof_r3_default.log 2>&1; echo "trace(default) rc=$?"
# (b) one 8-frame forward at a time (--depth 1 --launch-frames 0): the per-layer join, comparable with rounds 1-2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_r3 -o smap -- python $R/bench.py --depth 1 --launch-frames 0 --steps 6 --warmup 2 --no-cpu-baseline > $O/rocprof_r3.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_r3 -o pmc -- python $R/bench.py --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_r3.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_r3 -o pmc -- python $R/bench.py --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_r3.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_r3 -o pmc -- python $R/bench.py --depth 1 --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_mfma_r3.log 2>&1; echo "mfma rc=$?"
cd $R
DB=$(ls $O/prof_r3/*.db $O/prof_r3/*/*.db 2>/dev/null | head -1)
python tools/prof_export.py $DB $O/r3_final_x3_kernel_stats.csv
SMAP_PRECISION=x3 python tools/prof_layers.py $DB 8 > $O/r3_final_x3_layers.txt 2>&1; tail -3 $O/r3_final_x3_layers.txt
ST=$(ls $O/prof_r3_default/*kernel_stats.csv $O/prof_r3_default/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$ST" ] && cp $ST $O/r3_final_x3_rocprofv3_stats_native.csv
ST=$(ls $O/prof_r3/*kernel_stats.csv $O/prof_r3/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$ST" ] && cp $ST $O/r3_final_x3_rocprofv3_stats_native_8_frames_per_launch.csv
F=$(ls $O/pmc_fetch_r3/*counter_collection.csv $O/pmc_fetch_r3/*/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls $O/pmc_write_r3/*counter_collection.csv $O/pmc_write_r3/*/*counter_collection.csv 2>/dev/null | head -1)
python tools/prof_traffic.py $F $W $O/r3_final_hbm_traffic_x3.json | tail -12
python - <<'PY' | tee $O/r3_final_mfma_utilisation_pmc.log
import csv, glob, collections
f = (glob.glob('gpurun_out/pmc_mfma_r3/**/*counter_collection.csv', recursive=True) + glob.glob('gpurun_out/pmc_mfma_r3/*counter_collection.csv'))[0]
by = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    by[int(r['Dispatch_Id'])]['name'] = r['Kernel_Name']
ids = sorted(by)
stems = [i for i in ids if 'stem_kernel' in by[i]['name']]
seg = [i for i in ids if i >= stems[-1]]
heads = [i for i in seg if 'headsum' in by[i]['name']][:3]
seg = [i for i in seg if i <= heads[-1]]
conv = [i for i in seg if any(k in by[i]['name'] for k in ('conv_igemm', 'conv3x3_halo', 'convp_kernel'))]
mf = sum(by[i].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for i in conv)
ga = sum(by[i].get('GRBM_GUI_ACTIVE', 0) for i in conv)
print('x3 depth 1: launches', len(seg), 'conv', len(conv), 'MFMA busy cycles (sum over SIMDs) %.4g' % mf, 'GRBM_GUI_ACTIVE over conv kernels %.4g' % ga)
print('   (this rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs: %.4g / 8 = %.4g cycles = %.2f ms at 2.4 GHz, the serial conv time of one forward)' % (ga, ga / 8, ga / 8 / 2.4e6))
print('   MFMA pipe utilisation over the conv kernels = busy / (GUI_ACTIVE / 8 x 1024 SIMDs) = %.3f' % (mf / (ga / 8 * 1024)))
PY
rm -rf $O/prof_r3 $O/prof_r3_default $O/pmc_fetch_r3 $O/pmc_write_r3 $O/pmc_mfma_r3
