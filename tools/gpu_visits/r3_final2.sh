#!/bin/bash
# Round 3, after the final visit: the default line again (roofline.memory_path block added), the flip line WITH its parity block
# (the CPU child used a config without the mirror tables before), one cold B=1 line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py 2>&1 | tail -1 > $O/r3_final2_bench_x3.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3_final2_bench_x3.json'))
print(d['value'], json.dumps(d['roofline'].get('memory_path')), d['cpu_baseline']['value'])
PY
timeout 500 python bench.py --flip 2>&1 | tail -1 > $O/r3_final2_bench_x3_flip.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3_final2_bench_x3_flip.json'))
print(d['value'], json.dumps(d['config'].get('e2e_parity')), d.get('cpu_baseline', {}).get('sample', '')[:300])
PY
