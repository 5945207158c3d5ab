#!/bin/bash
# fused tail on the layer1 blocks (SMAP_TAIL=64:80) vs two launches: flip schedule (16 frames per launch), refine, depth 1
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f fps  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for spec in "" "64:80"; do
    echo -n "rep $rep SMAP_TAIL='$spec' --flip: "; SMAP_TAIL="$spec" run --flip
    echo -n "rep $rep SMAP_TAIL='$spec' --depth 1: "; SMAP_TAIL="$spec" run --depth 1
    echo -n "rep $rep SMAP_TAIL='$spec' --precision f16: "; SMAP_TAIL="$spec" run --precision f16
  done
done
