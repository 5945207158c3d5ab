#!/bin/bash
# Round 2, GPU visit 26: MFMA pipe utilisation of the whole forward from counters (x3 and f16, depth 1 so that kernels do not overlap)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for p in x3 f16; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc26_$p -o pmc -- python $R/bench.py --precision $p --depth 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc26_$p.log 2>&1; echo "rc=$? $p"
done
cd $R
python - <<'PY' | tee $O/r2_26_mfma_util.log
import csv, glob, collections
for p in ('x3', 'f16'):
    f = (glob.glob(f'gpurun_out/pmc26_{p}/**/*counter_collection.csv', recursive=True) + glob.glob(f'gpurun_out/pmc26_{p}/*counter_collection.csv'))[0]
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        by[int(r['Dispatch_Id'])]['name'] = r['Kernel_Name']
    ids = sorted(by)
    # last complete forward: from the last stem launch to the third headsum after it
    stems = [i for i in ids if 'stem_kernel' in by[i]['name']]
    last = stems[-1]
    seg = [i for i in ids if i >= last]
    heads = [i for i in seg if 'headsum' in by[i]['name']][:3]
    seg = [i for i in seg if i <= heads[-1]]
    conv = [i for i in seg if 'conv_igemm' in by[i]['name'] or 'conv3x3_halo' in by[i]['name']]
    mf = sum(by[i].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for i in conv)
    ga = sum(by[i].get('GRBM_GUI_ACTIVE', 0) for i in conv)
    gall = sum(by[i].get('GRBM_GUI_ACTIVE', 0) for i in seg)
    print(p, 'launches', len(seg), 'conv', len(conv), 'MFMA busy cycles (sum over SIMDs) %.4g' % mf, 'GRBM_GUI_ACTIVE conv %.4g all %.4g' % (ga, gall))
    for nsimd_div in (1024,):
        print('   MFMA pipe utilisation over the conv kernels = busy / (GUI_ACTIVE x %d SIMDs) = %.3f   (if GUI_ACTIVE is summed over 8 XCDs: %.3f)' % (nsimd_div, mf / (ga * nsimd_div), mf / (ga / 8 * nsimd_div)))
PY
rm -rf $O/pmc26_x3 $O/pmc26_f16
