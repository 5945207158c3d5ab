#!/bin/bash
# frames per launch: batch 8 (the BASELINE config) vs 16 and 4, depth 2 and 1 (x3)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f fps  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for b in 8 16 4; do
  for d in 2 1; do
    echo -n "--batch $b --depth $d: "; run --batch $b --depth $d --steps $((800 / b))
  done
done
echo -n "--batch 16 --depth 2 SMAP_MAX_FRAMES_PER_LAUNCH=8: "; SMAP_MAX_FRAMES_PER_LAUNCH=8 run --batch 16 --depth 2 --steps 50
