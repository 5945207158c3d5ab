#!/bin/bash
# Round 2, GPU visit 16: L2 hit rate / fetch bytes of single x3 layers (PMC, separate passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc16_$tag -o pmc -- python $R/tools/bench_conv.py --x3 --iters 4 --rotate 3 --only L2,L3,L4 --tile-override L2:52,L3:20,L4:21 > $O/pmc16_$tag.log 2>&1
  echo "rc=$? $c"
done
cd $R
python - <<'PY' | tee $O/r2_16_pmc_layers.log
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc16_*/**/*counter_collection.csv', recursive=True) + glob.glob('gpurun_out/pmc16_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'].split('<')[1].split('>')[0], r['Grid_Size'])
        rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, cs in sorted(rows.items()):
    print(key)
    for c, v in sorted(cs.items()):
        v = v[len(v)//2:]          # later launches (cold rotation steady)
        print('   %-28s mean %.4g  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
rm -rf $O/pmc16_*/
