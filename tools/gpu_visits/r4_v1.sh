#!/bin/bash
# round 4, visit 1: the host-side work of the round on hardware (tests, bench with timed-path parity, 16 vs 32 frames per launch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v1; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 400 python bench.py > $O/bench_x3.json 2> $O/bench_x3.err; echo "rc $?" >> $O/bench_x3.err
timeout 300 python bench.py --flip --steps 40 > $O/bench_x3_flip.json 2> $O/bench_x3_flip.err; echo "rc $?" >> $O/bench_x3_flip.err
for lf in 16 32 16 32; do
  timeout 300 python bench.py --launch-frames $lf --no-cpu-baseline --steps 60 2>>$O/ab_lf.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('launch-frames $lf', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/step', 'lf0', d['config'].get('value_launch_frames_0'), 'host', d['config']['host_ms_per_step'])
" >> $O/ab_lf.log
done
timeout 300 python bench.py --flip --launch-frames 32 --no-cpu-baseline --steps 40 2>>$O/ab_lf.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('flip launch-frames 32', round(d['value'],1),'fps', d['config']['frames_per_launch'])
" >> $O/ab_lf.log
cat $O/ab_lf.log
python -c "
import json
d=json.load(open('$O/bench_x3.json'))
print('bench', d['value'], d['config'].get('value_launch_frames_0'), json.dumps(d['config'].get('e2e_parity'))[:1500])
"
