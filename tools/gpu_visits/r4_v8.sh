#!/bin/bash
# round 4, visit 8: wave-state counters of the whole-Bottleneck kernels (single-op loop), two PMC passes
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=$R/gpurun_out/r4v8; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_a -o pmc -- python $R/tools/bench_convb.py 91 93 --n 5 > $O/pmc_a.log 2>&1
python $R/tools/prof_counters.py $(find $O/pmc_a -name "*counter_collection.csv" | head -1) bottleneck > $O/counters_a.txt 2>&1; cat $O/counters_a.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b -o pmc -- python $R/tools/bench_convb.py 91 93 --n 5 > $O/pmc_b.log 2>&1
python $R/tools/prof_counters.py $(find $O/pmc_b -name "*counter_collection.csv" | head -1) bottleneck > $O/counters_b.txt 2>&1; cat $O/counters_b.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_c -o pmc -- python $R/tools/bench_convb.py 91 93 --n 5 > $O/pmc_c.log 2>&1
python $R/tools/prof_counters.py $(find $O/pmc_c -name "*counter_collection.csv" | head -1) bottleneck > $O/counters_c.txt 2>&1; cat $O/counters_c.txt
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c
