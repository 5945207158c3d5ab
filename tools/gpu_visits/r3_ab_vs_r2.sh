#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/_r2 && python -m smap_amd.build > /dev/null 2>&1; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
for d in _r2 .; do
  for a in "--forward-only --batch 1 --steps 100 --warmup 10" "--depth 1 --steps 60 --warmup 10" "--steps 60 --warmup 10" "--steps 200 --warmup 10"; do
    v=$(cd $R/$d && timeout 300 python bench.py $a --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
    echo "$d | $a | $v"
  done
done
done
