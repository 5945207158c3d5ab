#!/bin/bash
# coalesced launches (two steps' batches per backbone launch): pipeline equality test, bench with and without
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_entry_gpu.py -q -x -k "two_stream_pipeline" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f fps  %.3f ms/step  fpl %s  mem-path %.0f GB/s  bb %.2f ms/launch' % (d['value'], d['ms_per_step'], d['config'].get('frames_per_launch'), d['roofline']['memory_path']['achieved_GBps'], d['roofline']['backbone_ms_per_launch']))"; }
for rep in 1 2; do
  echo -n "rep $rep default (16 frames per launch): "; run
  echo -n "rep $rep --launch-frames 0: "; run --launch-frames 0
done
echo -n "--depth 1: "; run --depth 1
echo -n "--depth 1 --launch-frames 0: "; run --depth 1 --launch-frames 0
echo -n "--refine: "; run --refine
echo -n "--precision f16: "; run --precision f16
echo -n "--precision f16 --launch-frames 0: "; run --precision f16 --launch-frames 0
echo -n "--flip: "; run --flip
echo -n "--steps 7 --warmup 3 (odd counts): "; run --steps 7 --warmup 3
