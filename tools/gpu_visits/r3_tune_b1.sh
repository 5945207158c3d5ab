#!/bin/bash
# Round 3: BASELINE configs[1] (batch 1, forward only) runs on heuristic tiles: tune its own table entries ("1,...").
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
SMAP_WPAIRS=1 timeout 1500 python tools/autotune.py --precision x3 --batch 1 --iters 30 --out $O/tile_table_x3_b1.json 2>&1 | grep -v amdgpu.ids | sed 's/{.*}//' | tail -42 | tee $O/r3_autotune_x3_b1.log
python - <<'PY'
import json
t = json.load(open('smap_amd/tile_table_x3.json'))
n = json.load(open('gpurun_out/tile_table_x3_b1.json'))
t.update({k: v for k, v in n.items() if k.startswith('1,')})
json.dump(t, open('gpurun_out/tile_table_x3_with_b1.json', 'w'), indent=0, sort_keys=True)
PY
for rep in 1 2; do
for tb in "" "$O/tile_table_x3_with_b1.json"; do
  v=$(SMAP_TILE_TABLE_X3=$tb timeout 300 python bench.py --forward-only --batch 1 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "b1 forward | table=${tb:-shipped} | $v" | tee -a $O/r3_ab_b1_table.log
  v=$(SMAP_TILE_TABLE_X3=$tb timeout 300 python bench.py --forward-only --batch 1 --graph --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "b1 forward graph | table=${tb:-shipped} | $v" | tee -a $O/r3_ab_b1_table.log
done
done
