cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backbone_gpu.py -q -x -k "fused_bottleneck_tail or fused_bottleneck_tails" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f fps  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  for spec in "" "64:80" "64:80,128:82"; do
    echo -n "rep $rep SMAP_TAIL='$spec' depth2: "; SMAP_TAIL="$spec" run
  done
done
bash tools/gpu_visits/r3_prof_tail.sh 2>&1 | grep tail | head -1
