#!/bin/bash
# GPU visit 3: re-run failing tests, conv micro-bench (tile variants), PMC counters on representative layers.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest gpu (ref + entry + assoc)" ; timeout 900 python -m pytest tests/test_ref_gpu.py tests/test_entry_gpu.py tests/test_assoc_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest_gpu2.log 2>&1 ; echo "pytest rc=$?" ; tail -15 $O/pytest_gpu2.log
echo "== microbench default tiles" ; timeout 300 python tools/bench_conv.py --iters 30 2>&1 | tee $O/mb_default.log
echo "== microbench overrides" ; timeout 300 python tools/bench_conv.py --iters 30 --tile-override L2:0,L6:0,L4:1,L1:1,L3:1,L7:1 2>&1 | tee $O/mb_override.log
timeout 300 python tools/bench_conv.py --iters 30 --only L2,L6,L5,L8 --tile-override L2:1,L6:1,L5:2,L8:2 2>&1 | tee -a $O/mb_override.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
pmc() { # tag, counters...
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/pmc_$tag -o pmc -- python $R/tools/bench_conv.py --iters 2 --only L1,L2,L3,L4 > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
pmc a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pmc b SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pmc c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc d FETCH_SIZE
pmc e WRITE_SIZE TCP_TCC_READ_REQ_sum
pmc f GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
ls $O | head -50
