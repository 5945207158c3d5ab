#!/bin/bash
# fused tail, phase-2 scheduling variants (tile 80 = plain, 83 = weight prefetch, 84 = barrier before stores, 86 = + residual ahead, 85 = all)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f fps  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for spec in "" "64:80" "64:83" "64:84" "64:86" "64:85"; do
    echo -n "rep $rep SMAP_TAIL='$spec' depth2: "; SMAP_TAIL="$spec" run
  done
done
