#!/bin/bash
# Round 2, GPU visit 1: split-precision (x3) kernels: parity (single conv, small schedule, end to end at B = 8),
# fp16 end-to-end measurement, first x3 bench numbers and a per-layer rocprofv3 trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q -p no:cacheprovider -k "split_precision" 2>&1 | tail -15 | tee $O/r2_01_tests_x3.log
timeout 1200 python -m pytest tests/test_e2e_parity_gpu.py -m gpu -q -p no:cacheprovider -s 2>&1 | tail -40 | tee $O/r2_01_tests_e2e.log
for p in x3 f16; do
  for d in 2 1; do
    echo "-- precision $p depth $d" | tee -a $O/r2_01_bench.log
    timeout 300 python bench.py --precision $p --depth $d --steps 16 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee -a $O/r2_01_bench.log
  done
done
cd /tmp; export TMPDIR=/tmp
SMAP_PRECISION=x3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_r2_01_x3 -o smap -- python $R/bench.py --precision x3 --depth 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/rocprof_r2_01_x3.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(ls $O/prof_r2_01_x3/*/*.db $O/prof_r2_01_x3/*.db 2>/dev/null | head -1)
echo "db=$DB"
if [ -n "$DB" ]; then
  python tools/prof_export.py $DB $O/r2_01_x3_kernel_stats.csv
  SMAP_PRECISION=x3 python tools/prof_layers.py $DB 8 > $O/r2_01_x3_layers.txt 2>&1; tail -5 $O/r2_01_x3_layers.txt
fi
