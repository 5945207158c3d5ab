#!/bin/bash
# round 4, visit 3: csrc/convb.hip with 16-channel stages + 8 KiB weight slots: phase stamps, probe, its tests, in-situ A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v3; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; x = torch.ones(8, device='cuda:0'); print('gpu ok', float(x.sum()))" > $O/sanity.log 2>&1; cat $O/sanity.log
timeout 300 python tools/debug/convb_probe.py > $O/probe.log 2>&1; echo "probe rc $?" >> $O/probe.log
grep -v "^   " $O/probe.log | head -20
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "whole_bottleneck or flip_tta_end_to_end or fused_bottleneck_tail" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
line() { python -c "
import sys,json
tag=sys.argv[1]
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(tag, round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/step', 'lf0', c.get('value_launch_frames_0') and round(c['value_launch_frames_0'],1), 'fpl', c['frames_per_launch'])
" "$1"; }
for rep in 1 2; do
  for blk in "" "64:90" "64:91"; do
    SMAP_BLOCK="$blk" timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep SMAP_BLOCK=[$blk] default" >> $O/ab_block.log
  done
done
for blk in "" "64:90" "64:91"; do
  SMAP_BLOCK="$blk" SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "SMAP_BLOCK=[$blk] depth1 lf0" >> $O/ab_block.log
done
cat $O/ab_block.log
