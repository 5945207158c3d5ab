#!/bin/bash
# GPU visit 14: PMC counters on the halo 3x3 kernel (layer3 / layer4 / layer2 shapes), 4- and 8-wave variants.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() { # tag tileset counters...
  tag=$1; tiles=$2; shift; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc3_$tag -o pmc -- python $R/tools/bench_conv.py --iters 2 --only L2,L6,L8 --tile-override $tiles > $O/pmc3_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
for v in "w4 L2:36,L6:36,L8:31" "w8 L2:57,L6:56,L8:51"; do
  set -- $v
  pmc a_$1 $2 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
  pmc b_$1 $2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA
  pmc c_$1 $2 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
done
ls $O/pmc3_a_w4
