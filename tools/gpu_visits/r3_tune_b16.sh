#!/bin/bash
# Round 3: the flip-TTA schedule runs 16 frames per launch (8 frames + their mirror images): tune its own table entries ("16,...").
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python tools/autotune.py --precision x3 --cold --batch 16 --iters 10 --out $O/tile_table_x3_b16.json 2>&1 | grep -v amdgpu.ids | sed 's/{.*}//' | tail -45 | tee $O/r3_autotune_x3_b16.log
python - <<'PY'
import json
t = json.load(open('smap_amd/tile_table_x3.json'))
n = json.load(open('gpurun_out/tile_table_x3_b16.json'))
t.update({k: v for k, v in n.items() if k.startswith('16,')})
json.dump(t, open('gpurun_out/tile_table_x3_with_b16.json', 'w'), indent=0, sort_keys=True)
print(len(t), 'entries')
PY
for rep in 1 2; do
for tb in "" "$O/tile_table_x3_with_b16.json"; do
  v=$(SMAP_TILE_TABLE_X3=$tb timeout 300 python bench.py --flip --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "flip | table=${tb:-shipped} | $v" | tee -a $O/r3_ab_flip_table.log
done
done
