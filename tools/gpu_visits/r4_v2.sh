#!/bin/bash
# round 4, visit 2: the whole suite on hardware (incl. csrc/convb.hip), the probe of the whole-Bottleneck kernel, the bench with
# timed-path parity, A/B of the whole-block launches and of 16 vs 32 frames per launch, a per-layer trace with them on.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v2; mkdir -p $O
export TMPDIR=/tmp
python -c "from smap_amd import lib; print(lib.version())" > $O/version.log 2>&1
timeout 300 python tools/debug/convb_probe.py > $O/probe.log 2>&1; echo "probe rc $?" >> $O/probe.log
cat $O/probe.log | head -60
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/step', 'lf0', c.get('value_launch_frames_0') and round(c['value_launch_frames_0'],1), 'fpl', c['frames_per_launch'], 'host', {k: (round(v,2) if isinstance(v,float) else v) for k,v in c['host_ms_per_step'].items() if k in ('submit_wall','enqueue_and_records','backpressure_wait')})
"; }
for rep in 1 2; do
  for blk in "" "64:90" "64:91"; do
    SMAP_BLOCK="$blk" timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep SMAP_BLOCK='$blk' default" >> $O/ab_block.log
  done
done
for blk in "" "64:90" "64:91"; do
  SMAP_BLOCK="$blk" SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "SMAP_BLOCK='$blk' depth1 lf0" >> $O/ab_block.log
done
cat $O/ab_block.log
for lf in 16 32 16 32; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --launch-frames $lf --no-cpu-baseline --steps 64 2>>$O/ab.err | line "launch-frames $lf" >> $O/ab_lf.log
done
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --flip --launch-frames 32 --no-cpu-baseline --steps 40 2>>$O/ab.err | line "flip launch-frames 32" >> $O/ab_lf.log
cat $O/ab_lf.log
timeout 400 python bench.py > $O/bench_x3.json 2> $O/bench_x3.err; echo "rc $?" >> $O/bench_x3.err
timeout 300 python bench.py --flip --steps 40 > $O/bench_x3_flip.json 2> $O/bench_x3_flip.err; echo "rc $?" >> $O/bench_x3_flip.err
python - <<'PY'
import json
for f in ("bench_x3", "bench_x3_flip"):
    try:
        d = json.load(open(f"gpurun_out/r4v2/{f}.json"))
        c = d["config"]
        print(f, round(d["value"], 1), "lf0", c.get("value_launch_frames_0"), "parity", json.dumps(c.get("e2e_parity"))[:900])
    except Exception as e:
        print(f, "ERR", e)
PY
# per-layer trace, depth 1, one launch per step, whole-block launches on (8 x 16 tiles) and off
for blk in "64:91" ""; do
  tag=$( [ -z "$blk" ] && echo off || echo on )
  SMAP_PRECISION=x3 SMAP_BLOCK="$blk" SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_$tag -o smap -- python bench.py --depth 1 --launch-frames 0 --steps 4 --warmup 2 --no-cpu-baseline > $O/rocprof_$tag.log 2>&1
  db=$(find $O/prof_$tag -name "*.db" | head -1)
  SMAP_PRECISION=x3 SMAP_BLOCK="$blk" python tools/prof_layers.py $db 8 > $O/layers_$tag.txt 2>&1
  python tools/prof_export.py $db $O/kernel_stats_$tag.csv
  rm -rf $O/prof_$tag
done
head -30 $O/layers_on.txt
