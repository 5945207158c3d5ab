#!/bin/bash
# Round 6: what each GPU visit ran, one function per visit (the logs under profiles/r6_v<N>_* came from these).
#     gpurun -- bash tools/gpu_visits/round6.sh <N>
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export TMPDIR=/tmp
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; r=d['roofline']; print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/step', 'lf0', c.get('value_launch_frames_0') and round(c['value_launch_frames_0'],1), 'fpl', c.get('frames_per_launch'), 'mfma', round(r['frac'],4), 'pipe', round(r.get('pipe_frac',0),4), 'assoc_us', c.get('association_lift_us_per_launch',{}).get('network'))
"; }
layers() {   # layers <out dir> <tag> <frames per launch> <bench args...>: per-op table of one rocprofv3 kernel trace
  local O=$1 tag=$2 B=$3; shift 3
  (cd /tmp; SMAP_PRECISION=x3 SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof_$tag -o smap -- python $R/bench.py "$@" --no-cpu-baseline > $R/$O/rocprof_$tag.log 2>&1)
  local db=$(find $R/$O/prof_$tag -name "*.db" | head -1)
  SMAP_PRECISION=x3 python tools/prof_layers.py $db $B > $O/layers_$tag.txt 2>&1
  python tools/prof_export.py $db $R/$O/kernel_stats_$tag.csv
  rm -rf $R/$O/prof_$tag
  tail -32 $O/layers_$tag.txt
}

v1() {
# visit 1: the round-5 tree on this round's box: driver-form line twice, the 100-step line, per-op traces at depth 1 with 16- and 8-frame launches
O=gpurun_out/r6v1; mkdir -p $O
python -c "from smap_amd import lib; print(lib.version())" > $O/version.log 2>&1
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>>$O/ab.err | line "rep $rep driver form" >> $O/base.log
done
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "100 steps" >> $O/base.log
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 --depth 1 2>>$O/ab.err | line "depth 1, 16 frames per launch" >> $O/base.log
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 --depth 1 --launch-frames 0 2>>$O/ab.err | line "depth 1, 8 frames per launch" >> $O/base.log
cat $O/base.log
layers $O d1_f16 16 --depth 1 --steps 6 --warmup 2
layers $O d1_f8 8 --depth 1 --launch-frames 0 --steps 4 --warmup 2
}

v2() {
# visit 2: convc.hip with the software pipeline (fragments of K step u + 1 in flight while step u multiplies; 2 residual chunks in registers):
# parity of tile 94, the launch alone (warm, back to back), in situ against round 5's loops and against 1 / 0 residual chunks in registers
O=gpurun_out/r6v2; mkdir -p $O
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "whole_bottleneck and (94 or 128)" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
V=$R/smap_amd/csrc/obj
for lib in "" $V/libsmap_hip_convc_convc_nrs3_convc_pipe0.so $V/libsmap_hip_convc_convc_nrs1_convc_pipe1.so $V/libsmap_hip_convc_convc_nrs0_convc_pipe1.so; do
  SMAP_HIP_LIB=$lib timeout 120 python tools/bench_convb.py 94 --n 40 --check 2>&1 | tail -2 >> $O/convc_alone.log
done
cat $O/convc_alone.log
for rep in 1 2; do
  for lib in "" $V/libsmap_hip_convc_convc_nrs3_convc_pipe0.so $V/libsmap_hip_convc_convc_nrs1_convc_pipe1.so $V/libsmap_hip_convc_convc_nrs0_convc_pipe1.so; do
    SMAP_HIP_LIB=$lib SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep depth 2 [$(basename ${lib:-shipped})]" >> $O/ab_convc.log
  done
done
for lib in "" $V/libsmap_hip_convc_convc_nrs3_convc_pipe0.so; do
  SMAP_HIP_LIB=$lib SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 --depth 1 2>>$O/ab.err | line "depth 1 [$(basename ${lib:-shipped})]" >> $O/ab_convc.log
done
cat $O/ab_convc.log
}

v3() {
# visit 3: the whole-block launches alone (warm, back to back): convc variants; trace of the default
O=gpurun_out/r6v3; mkdir -p $O
V=$R/smap_amd/csrc/obj
for lib in "" $V/libsmap_hip_convc_convc_nrs3_convc_pipe0.so $V/libsmap_hip_convc_convc_nrs1_convc_pipe1.so $V/libsmap_hip_convc_convc_nrs0_convc_pipe1.so; do
  SMAP_HIP_LIB=$lib timeout 120 python tools/bench_convb.py 94 --check 2>&1 | tail -2 >> $O/convc_alone.log
  SMAP_HIP_LIB=$lib timeout 120 python tools/bench_convb.py 94 2>&1 | tail -1 >> $O/convc_alone.log
done
cat $O/convc_alone.log
}

v4() {
# visit 4: the driver's form (20 steps, 5 warm-up) against pipeline depth and frames per launch
O=gpurun_out/r6v4; mkdir -p $O
for rep in 1 2; do
 for cfg in "--depth 2 --launch-frames 16" "--depth 3 --launch-frames 16" "--depth 3 --launch-frames 8" "--depth 4 --launch-frames 8" "--depth 2 --launch-frames 32"; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>>$O/ab.err | line "rep $rep 20 steps [$cfg]" >> $O/sweep.log
 done
done
for cfg in "--depth 2 --launch-frames 16" "--depth 3 --launch-frames 16" "--depth 3 --launch-frames 8"; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 $cfg 2>>$O/ab.err | line "100 steps [$cfg]" >> $O/sweep.log
done
cat $O/sweep.log
}

v5() {
# visit 5: the 256 x 256 register-epilogue tile (conv.hip, tile id 56): parity, then where it beats the table's tile (cold, alone), then in situ
O=gpurun_out/r6v5; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "x56 or t56 or x3-" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 900 python tools/autotune_regepi.py --batch 16 --batch 8 --iters 15 --out $R/$O/tile_table_x3_t56.json > $O/autotune_regepi.log 2>&1; tail -70 $O/autotune_regepi.log
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep shipped table" >> $O/ab_t56.log
  SMAP_TILE_TABLE_X3=$R/$O/tile_table_x3_t56.json SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep table with tile 56" >> $O/ab_t56.log
done
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 2>>$O/ab.err | line "depth 1 shipped table" >> $O/ab_t56.log
SMAP_TILE_TABLE_X3=$R/$O/tile_table_x3_t56.json SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 2>>$O/ab.err | line "depth 1 table with tile 56" >> $O/ab_t56.log
cat $O/ab_t56.log
}

v6() {
# visit 6: /255,/127 inside the head sum (smap_op.scale_hms), the peak search as two launches (smap_nms_ws), one result copy per association pass:
# association / entry / e2e tests, the bench line (association_lift_us_per_launch), the batch-1 full path
O=gpurun_out/r6v6; mkdir -p $O
timeout 1500 python -m pytest tests/test_assoc_gpu.py tests/test_ref_gpu.py tests/test_entry_gpu.py tests/test_e2e_parity_gpu.py tests/test_abi_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -12 $O/pytest.log
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>>$O/ab.err | line "rep $rep driver form" >> $O/bench.log
done
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "100 steps" >> $O/bench.log
timeout 300 python bench.py --no-cpu-baseline --batch 1 --depth 1 --launch-frames 0 --steps 200 --warmup 20 > $O/bench_b1_full_path.json 2>>$O/ab.err
python - <<'PY' >> gpurun_out/r6v6/bench.log
import json
d = json.load(open("gpurun_out/r6v6/bench_b1_full_path.json")); c = d["config"]
print("batch 1 full path, depth 1:", round(d["value"], 1), "fps", round(d["ms_per_step"], 3), "ms/frame", "assoc_us", c.get("association_lift_us_per_launch"))
PY
cat $O/bench.log
}

v7() {
# visit 7: the whole GPU suite on the round's changes so far (plan cache, eager lanes, per-op ticket memsets, scale_hms, nms_ws, rotation);
# split K with release / acquire on the ticket (-DSMAP_SPLITK_ACQREL=1) against the fence-free hand-off at batch 1; the bench line with 4 batches in rotation vs 1
O=gpurun_out/r6v7; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'], 'split-K launches', d['config'].get('split_k_launches'))
"; }
B1="--forward-only --batch 1 --steps 300 --warmup 30"
V=$R/smap_amd/csrc/obj/libsmap_hip_conv_splitk_acqrel1.so
for rep in 1 2; do
  timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 fence-free hand-off (shipped)" >> $O/ab_splitk.log
  SMAP_HIP_LIB=$V timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 release/acquire ticket" >> $O/ab_splitk.log
  SMAP_SPLITK=0 timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 no split K" >> $O/ab_splitk.log
done
SMAP_HIP_LIB=$V timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "split_k" > $O/pytest_acqrel.log 2>&1; echo "pytest rc $?" >> $O/pytest_acqrel.log
tail -3 $O/pytest_acqrel.log >> $O/ab_splitk.log
cat $O/ab_splitk.log
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 --rotate 4 2>>$O/ab.err | line "rep $rep 100 steps, 4 batches in rotation" >> $O/rotate.log
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 --rotate 1 2>>$O/ab.err | line "rep $rep 100 steps, the same batch every step" >> $O/rotate.log
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --rotate 4 2>>$O/ab.err | line "rep $rep 20 steps, 4 batches in rotation" >> $O/rotate.log
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --rotate 1 2>>$O/ab.err | line "rep $rep 20 steps, the same batch every step" >> $O/rotate.log
done
cat $O/rotate.log
}

v8() {
# visit 8: the whole GPU suite
O=gpurun_out/r6v8; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
}

v9() {
# visit 9: a stride-2 Bottleneck's last 1x1 + its shortcut conv as one K-concatenated GEMM (Graph.conv_cat, smap_op.in2_*): parity, then in situ
O=gpurun_out/r6v9; mkdir -p $O
timeout 1500 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "shortcut_conv_as_one_gemm or small_schedule or full_size or smap_module" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
  for v in "SMAP_CAT=0" "SMAP_CAT_TILE=50" "SMAP_CAT_TILE=51" "SMAP_CAT_TILE=20"; do
    env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep depth 2 [$v]" >> $O/ab_cat.log
  done
done
for v in "SMAP_CAT=0" "SMAP_CAT_TILE=50" "SMAP_CAT_TILE=51" "SMAP_CAT_TILE=20"; do
  env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 2>>$O/ab.err | line "depth 1 [$v]" >> $O/ab_cat.log
done
cat $O/ab_cat.log
layers $O d1_f16_cat 16 --depth 1 --steps 6 --warmup 2
}

v10() {
# visit 10: ResNet_top as one kernel (SMAP_STEMPOOL=1) in situ, once more on this round's schedule
O=gpurun_out/r6v10; mkdir -p $O
for rep in 1 2; do
  for v in "SMAP_X=0" "SMAP_STEMPOOL=1"; do
    env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep depth 2 [$v]" >> $O/ab.log
  done
done
cat $O/ab.log
}

v11() {
# visit 11: the root-depth head as tap dots + a stencil (Graph.conv_tapdot / tapsum): parity, in situ against round 5's three-way 1x1 + 3x3
O=gpurun_out/r6v11; mkdir -p $O
timeout 1500 python -m pytest tests/test_backbone_gpu.py tests/test_abi_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "tap_dots or small_schedule or full_size or smap_module or plan_blob or flip_tta or several_input" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
  for v in "SMAP_TAPHEAD=0" "SMAP_TAPHEAD=1"; do
    env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep depth 2 [$v]" >> $O/ab_tap.log
  done
done
for v in "SMAP_TAPHEAD=0" "SMAP_TAPHEAD=1"; do
  env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 2>>$O/ab.err | line "depth 1 [$v]" >> $O/ab_tap.log
done
cat $O/ab_tap.log
layers $O d1_f16_tap 16 --depth 1 --steps 6 --warmup 2
}

v12() {
# visit 12: tile of the two-way head 1x1 (N = 512 since the root-depth head left it), in situ
O=gpurun_out/r6v12; mkdir -p $O
for t in 20 50 51 54 56; do
  python - <<PY
import json
t = json.load(open("smap_amd/tile_table_x3.json"))
t["16,128,208,256,512,1,1"] = $t
json.dump(t, open("$O/table_$t.json", "w"), indent=0, sort_keys=True)
PY
done
for rep in 1 2; do
  for t in 20 50 51 54 56; do
    SMAP_TILE_TABLE_X3=$R/$O/table_$t.json SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep heads1x1 tile $t" >> $O/ab.log
  done
done
cat $O/ab.log
}

v13() {
# visit 13: skip1(x) + skip2(out) as one launch and one tensor (Graph.conv_relusum, smap_op.in2_mode = 1): parity, then in situ
O=gpurun_out/r6v13; mkdir -p $O
timeout 1500 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "two_activated_skip or small_schedule or full_size or smap_module or flip_tta" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
  for v in "SMAP_SKIPSUM=0" "SMAP_RELUSUM_TILE=50" "SMAP_RELUSUM_TILE=51"; do
    env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep depth 2 [$v]" >> $O/ab_skipsum.log
  done
done
for v in "SMAP_SKIPSUM=0" "SMAP_RELUSUM_TILE=50" "SMAP_RELUSUM_TILE=51"; do
  env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 2>>$O/ab.err | line "depth 1 [$v]" >> $O/ab_skipsum.log
done
cat $O/ab_skipsum.log
layers $O d1_f16_skipsum 16 --depth 1 --steps 6 --warmup 2
}

v14() {
# visit 14: wider tiles for the two-input launches (128 x 256 / 256 x 128 instances of conv_cat and conv_relusum), in situ
O=gpurun_out/r6v14; mkdir -p $O
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "two_activated_skip or shortcut_conv_as_one_gemm" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for rep in 1 2; do
  for v in "SMAP_RELUSUM_TILE=50" "SMAP_RELUSUM_TILE=54" "SMAP_RELUSUM_TILE=53" "SMAP_CAT_TILE=54" "SMAP_CAT_TILE=53" "SMAP_RELUSUM_TILE=54 SMAP_CAT_TILE=54"; do
    env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep depth 2 [$v]" >> $O/ab.log
  done
done
for v in "SMAP_RELUSUM_TILE=50" "SMAP_RELUSUM_TILE=54" "SMAP_RELUSUM_TILE=53" "SMAP_CAT_TILE=54"; do
  env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --depth 1 2>>$O/ab.err | line "depth 1 [$v]" >> $O/ab.log
done
cat $O/ab.log
}

v15() {
# visit 15: in situ, the 128 x 256 tile for the N >= 512 1x1 launches the cold isolated autotune gave 128 x 128 tiles
O=gpurun_out/r6v15; mkdir -p $O
python - <<PY
import json
base = json.load(open("smap_amd/tile_table_x3.json"))
var = {"l3c3": {"16,32,52,256,1024,1,1": 54}, "l4c3": {"16,16,26,512,2048,1,1": 54}, "l2c3cat_up3": {"16,64,104,256,512,1,1": 54},
       "l3l4": {"16,32,52,256,1024,1,1": 54, "16,16,26,512,2048,1,1": 54}, "l3c3_53": {"16,32,52,256,1024,1,1": 53}}
for k, v in var.items():
    t = dict(base); t.update(v)
    json.dump(t, open("$O/table_%s.json" % k, "w"), indent=0, sort_keys=True)
PY
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep shipped table" >> $O/ab.log
  for k in l3c3 l4c3 l2c3cat_up3 l3l4 l3c3_53; do
    SMAP_TILE_TABLE_X3=$R/$O/table_$k.json SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "rep $rep $k" >> $O/ab.log
  done
done
cat $O/ab.log
}

v16() {
# visit 16: batch 1 (configs[1]) with and without round 6's launches: the small-schedule rule
O=gpurun_out/r6v16; mkdir -p $O
b1() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1),'fps', round(d['ms_per_step'],3),'ms/frame', d['config']['launch'], 'conv launches', d['config'].get('conv_launches_per_forward'))
"; }
B1="--forward-only --batch 1 --steps 300 --warmup 30"
for rep in 1 2; do
  for v in "SMAP_X=0" "SMAP_TAPHEAD=1" "SMAP_CAT=1" "SMAP_SKIPSUM=1" "SMAP_CAT=1 SMAP_SKIPSUM=1 SMAP_TAPHEAD=1"; do
    env $v timeout 300 python bench.py $B1 2>>$O/ab.err | b1 "rep $rep b1 [$v]" >> $O/ab_b1.log
  done
done
env timeout 300 python bench.py $B1 --graph 2>>$O/ab.err | b1 "b1 rule, one HIP graph per forward" >> $O/ab_b1.log
cat $O/ab_b1.log
}

v17() {
# visit 17: are the two backbones in flight better off half a schedule apart?  (after a flush both are submitted back to back and run in step)
O=gpurun_out/r6v17; mkdir -p $O
for rep in 1 2; do
  for v in "SMAP_STAGGER_MS=0" "SMAP_STAGGER_MS=5" "SMAP_STAGGER_MS=9" "SMAP_STAGGER_MS=14"; do
    env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "rep $rep 100 steps [$v]" >> $O/ab.log
    env $v SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>>$O/ab.err | line "rep $rep 20 steps [$v]" >> $O/ab.log
  done
done
cat $O/ab.log
}

v18() {
# visit 18: peak search pass 3 with one lane per peak (49 loads in flight, the reference's accumulation order): bitwise vs the reference build, timing
O=gpurun_out/r6v18; mkdir -p $O
timeout 900 python -m pytest tests/test_assoc_gpu.py tests/test_ref_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
(cd /tmp; SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof -o smap -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/rocprof.log 2>&1)
db=$(find $R/$O/prof -name "*.db" | head -1); python tools/prof_export.py $db $R/$O/kernel_stats.csv; rm -rf $R/$O/prof
grep -i "nms\|group\|paf\|lift\|copyBuffer" $O/kernel_stats.csv | cut -c1-160
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 60 2>>$O/ab.err | line "60 steps" >> $O/bench.log
timeout 300 python bench.py --no-cpu-baseline --batch 1 --depth 1 --launch-frames 0 --steps 200 --warmup 20 2>>$O/ab.err | line "batch 1 full path" >> $O/bench.log
cat $O/bench.log
}

v19() {
# visit 19: where do the ~225 copyBuffer launches per step come from?  counts at 5 and 45 timed steps, and with the bench's synthetic extras off
O=gpurun_out/r6v19; mkdir -p $O
for cfg in "--steps 5 --warmup 5" "--steps 45 --warmup 5"; do
  (cd /tmp; SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/$O/prof -o smap -- python $R/bench.py $cfg --no-cpu-baseline > $R/$O/rocprof.log 2>&1)
  db=$(find $R/$O/prof -name "*.db" | head -1); python tools/prof_export.py $db $R/$O/ks.csv > /dev/null; rm -rf $R/$O/prof
  echo "== $cfg" >> $O/copies.log; grep -i "copyBuffer\|fillBuffer\|nms_mask\|lift_kernel\|stem_kernel" $O/ks.csv | cut -d, -f1-4 | cut -c1-90 >> $O/copies.log
done
cat $O/copies.log
}

v20() {
# visit 20: MFMA pipe utilisation by counters over the launches the DEFAULT pipeline issues (16 frames per launch, depth 1 so that kernels
# do not overlap), next to the 8-frame figure of validate_all.sh; and the driver's command three times (box-internal spread)
O=gpurun_out/r6v20; mkdir -p $O
(cd /tmp; SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_mfma -o pmc -- python $R/bench.py --depth 1 --steps 2 --warmup 2 --no-cpu-baseline > $R/$O/pmc_mfma.log 2>&1)
python tools/prof_mfma.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/mfma_utilisation_x3_16_frames.json 16 > $O/mfma_16.log 2>&1; cat $O/mfma_16.log
rm -rf $O/pmc_mfma
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_form_$i.json 2>> $O/bench.err
  python -c "import json; d = json.load(open('$O/driver_form_$i.json')); print('driver form run $i', round(d['value'], 1), d['ms_per_step'], d['config'].get('host_timeline_ms'))"
done
}

v21() {
# visit 21: what clock and power does the box hold under the bench?  (GRBM cycles of the PMC passes x 1/2.4 GHz are 4-14 % below the traced
# kernel durations: the shader clock under load is not the 2.4 GHz the peak is quoted at)
O=gpurun_out/r6v21; mkdir -p $O
rocm-smi --showclocks --showpower --showtemp > $O/idle.txt 2>&1
timeout 300 python bench.py --steps 1500 --warmup 10 --no-cpu-baseline > $O/bench_400.json 2> $O/bench.err &
BP=$!
i=0
while kill -0 $BP 2>/dev/null; do
  i=$((i+1)); echo "== sample $i $(date +%s.%N)" >> $O/load.txt
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\\|mclk\\|fclk\\|power\\|junction\\|edge" >> $O/load.txt
  sleep 0.3
done
wait $BP
python -c "import json; d = json.load(open('$O/bench_400.json')); print('bench 400 steps', round(d['value'], 1))"
grep -i "sclk\|power" $O/idle.txt | head -4
grep -i "sclk\|Power (W)" $O/load.txt | sort | uniq -c | sort -k1,1nr | head -40
}

v22() {
# visit 22: the bench line with the clock / power the box held in the timed region (benchkit/clocks.py)
O=gpurun_out/r6v22; mkdir -p $O
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/hwmon_files.txt 2>&1
for cfg in "--steps 20 --warmup 5" "" "--depth 1 --launch-frames 0 --steps 60" "--precision f16 --steps 60"; do
  n=$(echo "$cfg" | tr -d ' -'); timeout 400 python bench.py $cfg --no-cpu-baseline > $O/bench_$n.json 2>> $O/bench.err
  python -c "import json; d = json.load(open('$O/bench_$n.json')); r = d['roofline']; print('[$cfg]', round(d['value'], 1), 'fps; clocks', r.get('clocks'), 'pipe', round(r['pipe_frac'], 4), 'at clock', r.get('pipe_frac_at_measured_clock'), 'counters', r.get('pipe_frac_counters'), r.get('pipe_frac_counters_this_launch_size'))"
done
}

v23() {
# visit 23: split K with the release / acquire ticket as the shipped default: which K still gains at batch 1?  (and the split-K tests on it)
O=gpurun_out/r6v23; mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py -q -x -k "split_k" -p no:cacheprovider > $O/pytest_split_k.log 2>&1; tail -3 $O/pytest_split_k.log
for rep in 1 2; do
  for cfg in "SMAP_SPLITK_MINK=2048" "SMAP_SPLITK_MINK=2305" "SMAP_SPLITK_MINK=4608" "SMAP_SPLITK=0"; do
    env $cfg timeout 300 python bench.py --forward-only --batch 1 --graph --steps 300 --warmup 30 > $O/b1.json 2>> $O/bench.err
    python -c "import json; d = json.load(open('$O/b1.json')); print('rep $rep [$cfg] batch-1 forward graph', round(d['value'], 1), 'fps', round(d['ms_per_step'], 4), 'ms')" | tee -a $O/ab.log
  done
done
}

v24() {
# visit 24: the two headline lines again, now that the committed counter files carry this tree's build hash (validate_all.sh measures the
# counters AFTER its bench lines, which therefore print `traffic: null` whenever a hashed source changed since the previous passes)
O=gpurun_out/r6v24; mkdir -p $O
timeout 400 python bench.py > $O/bench_x3.json 2> $O/bench.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_x3_driver_form.json 2>> $O/bench.err
for f in bench_x3 bench_x3_driver_form; do
  python -c "import json; d = json.load(open('$O/$f.json')); r = d['roofline']; print('$f', round(d['value'], 1), 'frac', round(r['frac'], 4), 'traffic', r['traffic'], 'match', r['counters_match_build'], 'ctr', r['pipe_frac_counters'], r['pipe_frac_counters_this_launch_size'], 'clocks', r['clocks'] and (round(r['clocks']['sclk_mhz']), round(r['clocks']['power_w'] or 0)), 'at clock', r['pipe_frac_at_measured_clock'], 'cpu', d['cpu_baseline']['value'])"
done
}

v25() {
# visit 25: the fused bilinear add with its index arithmetic once per tile row (LDS records) instead of once per thread and pass:
# bit-for-bit digests of a seeded forward against the previous form (-DSMAP_EPI_ROWREC=0), the Upsample_unit tests, in situ A/B, per-op trace
O=gpurun_out/r6v25; mkdir -p $O
OLD=$R/smap_amd/csrc/obj/libsmap_hip_conv_epi_rowrec0.so
for cfg in "--batch 8 --precision x3" "--batch 1 --precision x3" "--batch 8 --precision f16"; do
  echo "== $cfg" >> $O/digest.log
  timeout 300 python tools/forward_digest.py $cfg 2>>$O/err.log | sed 's/^/new /' >> $O/digest.log
  SMAP_HIP_LIB=$OLD timeout 300 python tools/forward_digest.py $cfg 2>>$O/err.log | sed 's/^/old /' >> $O/digest.log
done
cat $O/digest.log
timeout 900 python -m pytest tests/test_backbone_gpu.py -q -x -k "upsample or bilinear or unit or small_schedule or merged" -p no:cacheprovider > $O/pytest_units.log 2>&1; tail -3 $O/pytest_units.log
for rep in 1 2; do
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "rep $rep records per row (new)" | tee -a $O/ab.log
  SMAP_HIP_LIB=$OLD SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "rep $rep per thread and pass (old)" | tee -a $O/ab.log
done
layers $O d1_f8_new 8 --depth 1 --launch-frames 0 --steps 4 --warmup 2 > /dev/null
grep "\.out \|total us" $O/layers_d1_f8_new.txt | cut -c1-150
}

v26() {
# visit 26: tile 57 (64 x 256, four waves of 64 x 64, 80 KiB of LDS: two workgroups per CU) under the fused bilinear add, where tile 54's
# 128 KiB epilogue tile leaves one workgroup per CU: the launch alone (cold operands), then in situ through alternative tile tables
O=gpurun_out/r6v26; mkdir -p $O
for t in 54 57 53; do
  timeout 200 python tools/bench_conv.py --x3 --rotate 3 --only L20,L21,L3 --tile-override L20:$t,L21:$t,L3:$t 2>>$O/err.log | tee -a $O/alone.log
done
for rep in 1 2; do
  for tb in "" tools/tables/x3_tile57_up4.json tools/tables/x3_tile57_up4_up3.json; do
    SMAP_TILE_TABLE_X3=$tb SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "rep $rep table [$tb]" | tee -a $O/ab.log
  done
done
}

v27() {
# visit 27: the whole library under other instruction-scheduling switches of the compiler (tools/build_flags.py): same bits? faster?
O=gpurun_out/r6v27; mkdir -p $O
timeout 300 python tools/forward_digest.py 2>>$O/err.log | sed 's/^/shipped /' | tee -a $O/digest.log
SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "shipped (first)" | tee -a $O/ab.log
for tag in maxilp maxmem nopostsched bias0 trackers; do
  LIB=$R/smap_amd/csrc/obj/libsmap_hip_flags_$tag.so
  SMAP_HIP_LIB=$LIB timeout 300 python tools/forward_digest.py 2>>$O/err.log | sed "s/^/$tag /" | tee -a $O/digest.log
  SMAP_HIP_LIB=$LIB SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "$tag" | tee -a $O/ab.log
  SMAP_BENCH_NO_LF0=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>>$O/ab.err | line "shipped (after $tag)" | tee -a $O/ab.log
done
}

v28() {
# visit 28: the image decoder hands over BGR bytes from PIL's own packer (no reversed-stride numpy copy) and copies once into page-locked
# memory: loader tests, then the shipped CLI end to end on 1024 image files
O=gpurun_out/r6v28; mkdir -p $O
timeout 1200 python -m pytest tests/test_entry_gpu.py -q -x -p no:cacheprovider > $O/pytest_entry.log 2>&1; tail -3 $O/pytest_entry.log
timeout 1500 python tools/cli_e2e.py --images 1024 --out $O/cli_e2e.json > $O/cli_e2e.log 2>&1
python -c "
import json
d = json.load(open('$O/cli_e2e.json'))
for r in d['runs']: print(r['case'], '|', round(r['frames_per_s'], 1), 'fps overall |', round(r['frames_per_s_after_engine_build'], 1), 'after engine build | loader', round(r['loader_s'], 2), 's submit', round(r['submit_s'], 2), 's')
"
}

v29() {
# visit 29: the CLI end to end again: process decoders, and the interpreter's switch interval
O=gpurun_out/r6v29; mkdir -p $O
timeout 1500 python tools/cli_e2e.py --images 1024 --out $O/cli_e2e.json > $O/cli_e2e.log 2>&1
python -c "
import json
d = json.load(open('$O/cli_e2e.json'))
for r in d['runs']: print(r['case'], '|', round(r['frames_per_s'], 1), 'fps overall |', round(r['frames_per_s_after_engine_build'], 1), 'after engine build | loader', round(r['loader_s'], 2), 's submit', round(r['submit_s'], 2), 's')
"
}

"v$1"
