#!/bin/bash
# Round 5: what each GPU visit ran, one function per visit (the logs under profiles/r5_v<N>_* came from these).
#     gpurun -- bash tools/gpu_visits/round5.sh <N>
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export TMPDIR=/tmp
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; r=d['roofline']; print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/step', 'lf0', c.get('value_launch_frames_0') and round(c['value_launch_frames_0'],1), 'fpl', c.get('frames_per_launch'), 'mfma', round(r['frac'],4), 'pipe', round(r.get('pipe_frac',0),4))
"; }

v1() {
# visit 1: the whole suite on hardware (merged 1x1 launches, RCCL one-rank gather, status words, arena budget, derived lifter-tie
# bounds), tile search for the merged launches, A/B merged vs one launch each, the bench line, a depth-1 per-layer trace
O=gpurun_out/r5v1; mkdir -p $O
python -c "from smap_amd import lib; print(lib.version())" > $O/version.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 600 python tools/autotune_seg.py --batch 8 --batch 16 --iters 15 --out $R/$O/tile_table_x3_seg.json > $O/autotune_seg.log 2>&1; tail -40 $O/autotune_seg.log
for rep in 1 2; do
  for m in 0 1; do
    SMAP_MERGE_1X1=$m SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep merge=$m shipped-table" >> $O/ab_merge.log
  done
  SMAP_TILE_TABLE_X3=$R/$O/tile_table_x3_seg.json SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep merge=1 tuned-table" >> $O/ab_merge.log
done
for m in 0 1; do
  SMAP_MERGE_1X1=$m SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "merge=$m depth1 lf0" >> $O/ab_merge.log
done
SMAP_TILE_TABLE_X3=$R/$O/tile_table_x3_seg.json SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "merge=1 tuned depth1 lf0" >> $O/ab_merge.log
cat $O/ab_merge.log
SMAP_TILE_TABLE_X3=$R/$O/tile_table_x3_seg.json timeout 400 python bench.py > $O/bench_x3.json 2> $O/bench_x3.err; echo "rc $?" >> $O/bench_x3.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5v1/bench_x3.json")); c = d["config"]
    print("bench", round(d["value"], 1), "lf0", c.get("value_launch_frames_0"), json.dumps(d["roofline"])[:600])
    print(json.dumps(c.get("e2e_parity"))[:1800])
except Exception as e:
    print("bench ERR", e)
PY
cd /tmp
SMAP_TILE_TABLE_X3=$R/$O/tile_table_x3_seg.json SMAP_PRECISION=x3 SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof_d1 -o smap -- python $R/bench.py --depth 1 --launch-frames 0 --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/rocprof_d1.log 2>&1
db=$(find $R/$O/prof_d1 -name "*.db" | head -1); (cd $R; SMAP_TILE_TABLE_X3=$R/$O/tile_table_x3_seg.json SMAP_PRECISION=x3 python tools/prof_layers.py $db 8 > $O/layers_d1.txt 2>&1); python $R/tools/prof_export.py $db $R/$O/kernel_stats_d1.csv; rm -rf $R/$O/prof_d1
tail -25 $R/$O/layers_d1.txt
}

v2() {
# visit 2: split-K / merge tests; merge policy A/B (0 = one launch per conv, 1 = the merges that pay, 2 = all); the 4x16 whole-block
# tiles with THREE workgroups per CU (diagnostics build) against the shipped 8x16 tiles; batch 1: split K off / rule / forced, graph form,
# per-layer trace; the shipped CLI end to end on a generated image folder
O=gpurun_out/r5v2; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "split_k or merged" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
  for m in 0 1 2; do
    SMAP_MERGE_1X1=$m SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep merge=$m" >> $O/ab_merge.log
  done
done
cat $O/ab_merge.log
V3=$R/smap_amd/csrc/obj/libsmap_hip_convb_convb_wgs43_convb_lds4_kb52.so
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep 8x16 tiles (shipped)" >> $O/ab_convb.log
  SMAP_BLOCK="64:90,128:94" SMAP_BLOCK_FIRST="64:92" SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep 4x16 tiles, 2 workgroups per CU" >> $O/ab_convb.log
  SMAP_HIP_LIB=$V3 SMAP_BLOCK="64:90,128:94" SMAP_BLOCK_FIRST="64:92" SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep 4x16 tiles, 3 workgroups per CU" >> $O/ab_convb.log
done
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "depth1 lf0 8x16 tiles (shipped)" >> $O/ab_convb.log
SMAP_HIP_LIB=$V3 SMAP_BLOCK="64:90,128:94" SMAP_BLOCK_FIRST="64:92" SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "depth1 lf0 4x16 tiles, 3 workgroups per CU" >> $O/ab_convb.log
cat $O/ab_convb.log
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'])
"; }
for sk in 0 1 2 4 8; do
  SMAP_SPLITK=$sk timeout 300 python bench.py --forward-only --batch 1 --steps 300 --warmup 30 2>>$O/ab.err | b1 "b1 SMAP_SPLITK=$sk" >> $O/ab_b1.log
done
SMAP_SPLITK=0 timeout 300 python bench.py --forward-only --batch 1 --steps 300 --warmup 30 --graph 2>>$O/ab.err | b1 "b1 SMAP_SPLITK=0 graph" >> $O/ab_b1.log
timeout 300 python bench.py --forward-only --batch 1 --steps 300 --warmup 30 --graph 2>>$O/ab.err | b1 "b1 rule graph" >> $O/ab_b1.log
SMAP_MERGE_1X1=0 SMAP_SPLITK=0 timeout 300 python bench.py --forward-only --batch 1 --steps 300 --warmup 30 2>>$O/ab.err | b1 "b1 round-4 schedule (no merges, no split K)" >> $O/ab_b1.log
cat $O/ab_b1.log
cd /tmp
for sk in 0 1; do
  SMAP_SPLITK=$sk SMAP_PRECISION=x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof_b1_$sk -o smap -- python $R/bench.py --forward-only --batch 1 --steps 20 --warmup 5 > $R/$O/rocprof_b1_$sk.log 2>&1
  db=$(find $R/$O/prof_b1_$sk -name "*.db" | head -1); (cd $R; SMAP_SPLITK=$sk SMAP_PRECISION=x3 python tools/prof_layers.py $db 1 > $O/layers_b1_splitk$sk.txt 2>&1); rm -rf $R/$O/prof_b1_$sk
  tail -22 $R/$O/layers_b1_splitk$sk.txt
done
cd $R
timeout 1500 python tools/cli_e2e.py --images 256 --out $R/$O/cli_e2e.json > $O/cli_e2e.log 2>&1; tail -8 $O/cli_e2e.log
}

v3() {
# visit 3: split K without fences (sc1 accesses), lanes, small schedules without layer2's whole-block launches: tests, batch-1 sweeps and
# per-layer trace; the staggered start of the 8x16 whole-block tiles in situ; the CLI end to end on 1024 images with skeletons
O=gpurun_out/r5v3; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "split_k or merged or lanes or graph_replay or overlapping" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 900 python -m pytest tests/test_entry_gpu.py tests/test_abi_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_entry.log 2>&1; echo "pytest rc $?" >> $O/pytest_entry.log
tail -6 $O/pytest_entry.log
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'])
"; }
B1="--forward-only --batch 1 --steps 300 --warmup 30"
SMAP_SPLITK=0 SMAP_LANES=0 SMAP_BLOCK="64:91,128:94" SMAP_MERGE_1X1=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 round-4 schedule" >> $O/ab_b1.log
SMAP_SPLITK=0 SMAP_LANES=0 SMAP_BLOCK="64:91,128:94" timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 + merged 1x1" >> $O/ab_b1.log
SMAP_SPLITK=0 SMAP_LANES=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 + layer2 blocks as three launches" >> $O/ab_b1.log
SMAP_SPLITK=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 + lanes" >> $O/ab_b1.log
SMAP_LANES=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 + split K (rule t256), no lanes" >> $O/ab_b1.log
timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 default (split K t256 + lanes)" >> $O/ab_b1.log
SMAP_SPLITK=t384 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 split K t384 + lanes" >> $O/ab_b1.log
SMAP_SPLITK=t512 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 split K t512 + lanes" >> $O/ab_b1.log
timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "b1 default, graph" >> $O/ab_b1.log
SMAP_LANES=0 timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "b1 default without lanes, graph" >> $O/ab_b1.log
cat $O/ab_b1.log
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep shipped" >> $O/ab_stagger.log
  for us in 10 20; do
    SMAP_HIP_LIB=$R/smap_amd/csrc/obj/libsmap_hip_convb_convb_stagger_us$us.so SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep layer1 blocks: second slots start $us us late" >> $O/ab_stagger.log
  done
done
for lib in "" $R/smap_amd/csrc/obj/libsmap_hip_convb_convb_stagger_us10.so $R/smap_amd/csrc/obj/libsmap_hip_convb_convb_stagger_us20.so; do
  SMAP_HIP_LIB=$lib SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "depth1 lf0 lib=$(basename "$lib")" >> $O/ab_stagger.log
done
cat $O/ab_stagger.log
cd /tmp
SMAP_PRECISION=x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof_b1 -o smap -- python $R/bench.py --forward-only --batch 1 --steps 20 --warmup 5 > $R/$O/rocprof_b1.log 2>&1
db=$(find $R/$O/prof_b1 -name "*.db" | head -1); (cd $R; SMAP_PRECISION=x3 python tools/prof_layers.py $db 1 > $O/layers_b1.txt 2>&1); rm -rf $R/$O/prof_b1
tail -22 $R/$O/layers_b1.txt
cd $R
timeout 1500 python tools/cli_e2e.py --images 1024 --out $R/$O/cli_e2e.json > $O/cli_e2e.log 2>&1; tail -8 $O/cli_e2e.log | cut -c1-600
}

v4() {
# visit 4: the 8x16 whole-block tiles of layer1 with a staggered start of the CUs' second slots and / or raised issue priority during MFMA
# groups (diagnostics builds), in situ and at depth 1; batch 1 with the final split-K rule
O=gpurun_out/r5v4; mkdir -p $O
OBJ=$R/smap_amd/csrc/obj
for lib in "" $OBJ/libsmap_hip_convb_convb_stagger_us10.so $OBJ/libsmap_hip_convb_convb_stagger_us20.so $OBJ/libsmap_hip_convb_convb_setprio1.so $OBJ/libsmap_hip_convb_convb_setprio1_convb_stagger_us20.so ""; do
  SMAP_HIP_LIB=$lib SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "depth2 lib=$(basename "$lib")" >> $O/ab_convb.log
done
for lib in "" $OBJ/libsmap_hip_convb_convb_stagger_us10.so $OBJ/libsmap_hip_convb_convb_stagger_us20.so $OBJ/libsmap_hip_convb_convb_setprio1.so $OBJ/libsmap_hip_convb_convb_setprio1_convb_stagger_us20.so; do
  SMAP_HIP_LIB=$lib SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 --launch-frames 0 2>>$O/ab.err | line "depth1 lf0 lib=$(basename "$lib")" >> $O/ab_convb.log
done
cat $O/ab_convb.log; tail -3 $O/ab.err
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'])
"; }
B1="--forward-only --batch 1 --steps 300 --warmup 30"
SMAP_SPLITK=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 no split K" >> $O/ab_b1.log
timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 default (split K where K >= 2048)" >> $O/ab_b1.log
timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "b1 default, graph" >> $O/ab_b1.log
SMAP_SPLITK=0 timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "b1 no split K, graph" >> $O/ab_b1.log
cat $O/ab_b1.log
}

v5() {
# visit 5: the entry / ABI tests with the summation order pinned (split K off in that file), the CLI end to end with the final loader
O=gpurun_out/r5v5; mkdir -p $O
timeout 900 python -m pytest tests/test_entry_gpu.py tests/test_abi_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_entry.log 2>&1; echo "pytest rc $?" >> $O/pytest_entry.log
tail -6 $O/pytest_entry.log
timeout 1500 python tools/cli_e2e.py --images 1024 --out $R/$O/cli_e2e.json > $O/cli_e2e.log 2>&1; tail -8 $O/cli_e2e.log | cut -c1-700
}

v6() {
# the round-end validation (tools/gpu_visits/validate_all.sh): whole suite, smoke, every bench line, traces, PMC passes, host budget, CLI
bash tools/gpu_visits/validate_all.sh r5final
}

v7() {
# visit 7: the split-K stress test (300 batch-1 forwards on two streams, bit for bit) and the two trimmed tests
O=gpurun_out/r5v7; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_entry_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "many_runs or split_k or cli_device_preprocess" --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -22 $O/pytest.log
}

v8() {
# visit 8: what the driver runs at round end, verbatim, on a fresh box: the suite with -x, smoke, the bench with its short form
O=gpurun_out/r5v8; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/bench_driver_form.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config'].get('conv_launches_per_forward'), d['cpu_baseline']['value'])"
}

v9() {
# visit 9: tile 7 (64x64, three 64-half K tiles in flight, 128 KiB) for the batch-1 schedule: parity cases, then batch 1 with every
# tile-2 launch remapped to it, kernel by kernel, as a graph, and per layer
O=gpurun_out/r5v9; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "x1x7] or x2x7] or x1x7 or tile7" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'])
"; }
B1="--forward-only --batch 1 --steps 300 --warmup 30"
for rep in 1 2; do
  timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 tile 2" >> $O/ab_b1.log
  SMAP_TILE_REMAP=2:7 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 tile 2 -> 7" >> $O/ab_b1.log
done
timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "b1 tile 2, graph" >> $O/ab_b1.log
SMAP_TILE_REMAP=2:7 timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "b1 tile 2 -> 7, graph" >> $O/ab_b1.log
SMAP_TILE_REMAP=2:7 SMAP_SPLITK=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 tile 2 -> 7, no split K" >> $O/ab_b1.log
SMAP_TILE_REMAP=2:7 SMAP_SPLITK=t512 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 tile 2 -> 7, split K t512 any K" >> $O/ab_b1.log
cat $O/ab_b1.log; tail -3 $O/ab.err
cd /tmp
for rm in "" "2:7"; do
  SMAP_TILE_REMAP=$rm SMAP_PRECISION=x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof_b1 -o smap -- python $R/bench.py --forward-only --batch 1 --steps 20 --warmup 5 > $R/$O/rocprof_b1.log 2>&1
  db=$(find $R/$O/prof_b1 -name "*.db" | head -1); (cd $R; SMAP_TILE_REMAP=$rm SMAP_PRECISION=x3 python tools/prof_layers.py $db 1 > $O/layers_b1_remap_$(echo $rm | tr ':' '_').txt 2>&1); rm -rf $R/$O/prof_b1
done
tail -22 $R/$O/layers_b1_remap_2_7.txt
}

v10() {
# visit 10: the deep-pipeline rule (tile 2 -> 7 for launches of <= 256 workgroups) as shipped: stress test + smoke, batch 1 kernel by
# kernel / as a graph / per layer
O=gpurun_out/r5v10; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "many_runs or full_size_forward_vs_imported" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'], d['config'].get('split_k_launches'))
"; }
B1="--forward-only --batch 1 --steps 300 --warmup 30"
for rep in 1 2; do
  SMAP_DEEP_TILE=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 tile 2 everywhere" >> $O/ab_b1.log
  timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 shipped rule" >> $O/ab_b1.log
  timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "rep $rep b1 shipped rule, graph" >> $O/ab_b1.log
done
cat $O/ab_b1.log; tail -3 $O/ab.err
timeout 300 python bench.py $B1 --graph > $O/bench_x3_forward_b1.json 2>>$O/ab.err
cd /tmp
SMAP_PRECISION=x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof_b1 -o smap -- python $R/bench.py --forward-only --batch 1 --steps 20 --warmup 5 > $R/$O/rocprof_b1.log 2>&1
db=$(find $R/$O/prof_b1 -name "*.db" | head -1); (cd $R; SMAP_PRECISION=x3 python tools/prof_layers.py $db 1 > $O/layers_b1.txt 2>&1); rm -rf $R/$O/prof_b1
tail -20 $R/$O/layers_b1.txt
}

v11() {
# visit 11: batch 1 with layer1's whole-block launches on 4x16 tiles (416 workgroups of half the work instead of 208)
O=gpurun_out/r5v11; mkdir -p $O
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'])
"; }
B1="--forward-only --batch 1 --steps 300 --warmup 30"
for rep in 1 2; do
  timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "rep $rep b1 8x16 whole-block tiles, graph" >> $O/ab_b1.log
  SMAP_BLOCK="64:90" SMAP_BLOCK_FIRST="64:92" timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "rep $rep b1 4x16 whole-block tiles, graph" >> $O/ab_b1.log
  SMAP_BLOCK="64:90" SMAP_BLOCK_FIRST="64:93" timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "rep $rep b1 4x16 identity blocks, 8x16 first blocks, graph" >> $O/ab_b1.log
  SMAP_BLOCK="" SMAP_BLOCK_FIRST="" timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "rep $rep b1 layer1 as three launches per block, graph" >> $O/ab_b1.log
done
SMAP_BLOCK="64:90" SMAP_BLOCK_FIRST="64:92" timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "b1 4x16 whole-block tiles, kernel by kernel" >> $O/ab_b1.log
cat $O/ab_b1.log; tail -3 $O/ab.err
}

v12() {
# visit 12: the final batch-1 schedule (4x16 whole-block tiles in small schedules) -- parity at full size, stress, entry tests, smoke, bench lines
O=gpurun_out/r5v12; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "many_runs or full_size or lanes or graph_replay" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python -m pytest tests/test_entry_gpu.py tests/test_abi_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_entry.log 2>&1; echo "pytest rc $?" >> $O/pytest_entry.log; tail -4 $O/pytest_entry.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python bench.py --forward-only --batch 1 --graph --steps 300 --warmup 30 > $O/bench_x3_forward_b1.json 2>>$O/ab.err
timeout 300 python bench.py --forward-only --batch 1 --steps 300 --warmup 30 > $O/bench_x3_forward_b1_kernel_by_kernel.json 2>>$O/ab.err
python -c "
import json
for f in ('bench_x3_forward_b1','bench_x3_forward_b1_kernel_by_kernel'):
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['value'],1), round(d['ms_per_step'],3), d['config'])"
cd /tmp
SMAP_PRECISION=x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof_b1 -o smap -- python $R/bench.py --forward-only --batch 1 --steps 20 --warmup 5 > $R/$O/rocprof_b1.log 2>&1
db=$(find $R/$O/prof_b1 -name "*.db" | head -1); (cd $R; SMAP_PRECISION=x3 python tools/prof_layers.py $db 1 > $O/layers_b1.txt 2>&1); rm -rf $R/$O/prof_b1
tail -18 $R/$O/layers_b1.txt
}

v13() {
# visit 13: the part of the suite that visit 8b did not reach (-x stopped at a test whose expectation the deep-pipeline rule had changed):
# test_backbone_gpu.py from the split-K tests on, the end-to-end parity file, the reference-build file
O=gpurun_out/r5v13; mkdir -p $O
python -m pytest tests/test_backbone_gpu.py --collect-only -q -m gpu 2>/dev/null | grep "::" | sed -n '219,400p' > $O/ids.txt
timeout 900 python -m pytest $(cat $O/ids.txt) tests/test_e2e_parity_gpu.py tests/test_ref_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
}

v14() {
# visit 14: the split-K schedule tests after the test fix of visit 13 (the one the driver's -x run would have stopped at)
O=gpurun_out/r5v14; mkdir -p $O
timeout 300 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "small_schedule_split_k" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
}

"v$1"
