#!/bin/bash
# Round 3: fused Bottleneck tails (csrc/convf.hip) in situ -- default schedule vs SMAP_TAIL variants, same box, interleaved twice.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f fps  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for spec in "" "64:80" "64:81" "64:80,128:82"; do
    echo -n "rep $rep SMAP_TAIL='$spec' depth2: "; SMAP_TAIL="$spec" run
  done
done
for spec in "" "64:80" "64:80,128:82"; do
  echo -n "SMAP_TAIL='$spec' depth1: "; SMAP_TAIL="$spec" run --depth 1
  echo -n "SMAP_TAIL='$spec' B=1 forward-only: "; SMAP_TAIL="$spec" run --forward-only --batch 1
done
