#!/bin/bash
# Round 4: what each GPU visit ran, one function per visit (the logs under profiles/r4_v<N>_* came from these).
#     gpurun -- bash tools/gpu_visits/round4.sh <N>
# Generic, parameterised scripts: ab_bench.sh (interleaved in-situ A/B of env settings), gpu_profile.sh (rocprofv3 trace + PMC passes),
# validate_all.sh (the round-end validation).

v1() {
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
}

v2() {
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
}

v3() {
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
}

v4() {
# round 4, visit 4: the first-block variant of csrc/convb.hip (tile ids 92, 93): probe, tests, in-situ A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v4; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/debug/convb_probe.py 92 93 > $O/probe.log 2>&1; echo "probe rc $?" >> $O/probe.log
grep -v "^   " $O/probe.log | head -20
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "whole_bottleneck or fused_bottleneck_tail" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
bash tools/gpu_visits/ab_bench.sh $O/ab_block.log 2 "--no-cpu-baseline --steps 60" "" "SMAP_BLOCK=64:91" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:92" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93 SMAP_LAUNCH_FRAMES_X=1" > /dev/null
bash tools/gpu_visits/ab_bench.sh $O/ab_block_d1.log 1 "--no-cpu-baseline --steps 40 --depth 1 --launch-frames 0" "" "SMAP_BLOCK=64:91" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:92" > /dev/null
bash tools/gpu_visits/ab_bench.sh $O/ab_block_lf32.log 1 "--no-cpu-baseline --steps 64 --launch-frames 32" "" "SMAP_BLOCK=64:91 SMAP_BLOCK_FIRST=64:93" > /dev/null
cat $O/ab_block.log $O/ab_block_d1.log $O/ab_block_lf32.log
}

v6() {
# round 4, visit 6: decomposition of the whole-Bottleneck launch by ablation builds; the three re-scoped tests; --refine parity
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v6; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/bench_convb.py 91 90 93 92 > $O/convb_ablation.log 2>&1
for n in 1 2 4 8 3 6 9 15; do
  SMAP_HIP_LIB=$PWD/smap_amd/csrc/obj/libsmap_hip_convb$n.so timeout 200 python tools/bench_convb.py 91 90 >> $O/convb_ablation.log 2>&1
done
grep "us per launch" $O/convb_ablation.log
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "split_precision_every_tensor or fused_bottleneck_tails" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 400 python bench.py --refine --steps 60 > $O/bench_x3_refine.json 2> $O/bench_x3_refine.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4v6/bench_x3_refine.json")); c = d["config"]; m = c["e2e_parity"]
print("refine", round(d["value"], 1), {k: m.get(k) for k in ("peaks_differing", "peaks_clear_mismatch", "max_joint_err_cm", "joints_over_0.1cm_unexplained", "timed_records_equal_these_frames")})
PY
}

v7() {
# round 4, visit 7: does the whole-Bottleneck kernel run two workgroups per CU?  occupancy query + 64 / 72 KiB LDS variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v7; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/bench_convb.py 91 90 93 92 > $O/convb_lds.log 2>&1
for n in lds64 lds72; do
  SMAP_HIP_LIB=$PWD/smap_amd/csrc/obj/libsmap_hip_convb$n.so timeout 200 python tools/bench_convb.py 91 90 93 92 >> $O/convb_lds.log 2>&1
done
grep "us per launch" $O/convb_lds.log
}

v8() {
# round 4, visit 8: wave-state counters of the whole-Bottleneck kernels (single-op loop), two PMC passes
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=$R/gpurun_out/r4v8; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_a -o pmc -- python $R/tools/bench_convb.py 91 93 --n 5 > $O/pmc_a.log 2>&1
python $R/tools/prof_counters.py $(find $O/pmc_a -name "*counter_collection.csv" | head -1) bottleneck > $O/counters_a.txt 2>&1; cat $O/counters_a.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b -o pmc -- python $R/tools/bench_convb.py 91 93 --n 5 > $O/pmc_b.log 2>&1
python $R/tools/prof_counters.py $(find $O/pmc_b -name "*counter_collection.csv" | head -1) bottleneck > $O/counters_b.txt 2>&1; cat $O/counters_b.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_c -o pmc -- python $R/tools/bench_convb.py 91 93 --n 5 > $O/pmc_c.log 2>&1
python $R/tools/prof_counters.py $(find $O/pmc_c -name "*counter_collection.csv" | head -1) bottleneck > $O/counters_c.txt 2>&1; cat $O/counters_c.txt
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c
}

v9() {
# round 4, visit 9: cheap in-situ A/Bs on the new default schedule: fused stem + max-pool, three backbones in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v9; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_visits/ab_bench.sh $O/ab_misc.log 2 "--no-cpu-baseline --steps 60" "" "SMAP_STEMPOOL=1" > /dev/null
bash tools/gpu_visits/ab_bench.sh $O/ab_depth3.log 2 "--no-cpu-baseline --steps 60 --depth 3" "" > /dev/null
cat $O/ab_misc.log $O/ab_depth3.log
}

v10() {
# round 4, visit 10: in-situ coordinate descent over the tiles of the heaviest 16-frame shapes on the new schedule (whole-block launches on)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v10; mkdir -p $O
export TMPDIR=/tmp SMAP_BENCH_NO_LF0=1
timeout 840 python tools/insitu_tune.py --precision x3 --steps 40 --warmup 6 --candidates tools/insitu_candidates_r4.json --out $O/tile_table_x3_insitu.json > $O/insitu.log 2>&1
cat $O/insitu.log
}

v11() {
# round 4, visit 11: eight-wave halo 3x3 tiles (40..43): parity of the new instances, then cold isolated timings against the shipped tiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v11; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_backbone_gpu.py -q -x -m gpu -k "x1x40 or x1x41 or x1x42 or x1x43 or halo_conv" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 240 python tools/bench_halo8.py --batch 16 > $O/halo8_b16.log 2>&1; cat $O/halo8_b16.log
timeout 200 python tools/bench_halo8.py --batch 8 > $O/halo8_b8.log 2>&1; cat $O/halo8_b8.log
}

v12() {
# round 4, visit 12: staggered halo 3x3 schedule (tiles 44 / 45): parity, race screen against the lockstep tiles, cold timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v12; mkdir -p $O
export TMPDIR=/tmp
timeout 180 python -m pytest tests/test_backbone_gpu.py -q -x -m gpu -k "x1x44 or x1x45" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
grep -q passed $O/pytest.log || exit 1
timeout 240 python tools/debug/halo_stag_race.py --runs 30 > $O/race_x3.log 2>&1; tail -12 $O/race_x3.log
timeout 120 python tools/debug/halo_stag_race.py --runs 10 --precision f16 > $O/race_f16.log 2>&1; tail -3 $O/race_f16.log
timeout 240 python tools/bench_halo8.py --batch 16 > $O/halo8_b16.log 2>&1; cat $O/halo8_b16.log
timeout 200 python tools/bench_halo8.py --batch 8 > $O/halo8_b8.log 2>&1; cat $O/halo8_b8.log
}

v13() {
# round 4, visit 13: in-situ A/B of the table with the staggered eight-wave halo tiles (44 / 45) against the previous table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v13; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_visits/ab_bench.sh $O/ab_halo8.log 3 "--no-cpu-baseline --steps 40 --warmup 6" "SMAP_TILE_TABLE_X3=tools/tile_table_x3_r3.json" ""
}

v15() {
# visit 15: issue interval of the matrix instructions per wave (hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_issue.hip -o tools/ubench/mfma_issue first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v15; mkdir -p $O
timeout 120 tools/ubench/mfma_issue > $O/mfma_issue.log 2>&1; cat $O/mfma_issue.log
}

v16() {
# visit 16: ablation builds of the staggered halo loop (python tools/build_ablate.py --conv3 1 2 16 8 3 18 19 51 27 59 first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v16; mkdir -p $O; export TMPDIR=/tmp
for n in 0 1 2 16 8 3 18 19 51 27 59; do
  lib=$PWD/smap_amd/csrc/obj/libsmap_hip_conv3abl$n.so; [ $n = 0 ] && lib=$PWD/smap_amd/libsmap_hip.so
  echo "ablate $n" >> $O/conv3_ablation.log
  SMAP_HIP_LIB=$lib timeout 100 python tools/bench_halo8.py --batch 16 --only 32,52,256,256 --tiles 45,43 2>&1 | grep -v amdgpu.ids >> $O/conv3_ablation.log
  SMAP_HIP_LIB=$lib timeout 100 python tools/bench_halo8.py --batch 16 --only 64,104,128,128 --tiles 44,31 2>&1 | grep -v amdgpu.ids >> $O/conv3_ablation.log
done; cat $O/conv3_ablation.log
}

v18() {
# visits 17, 18: LDS-DMA requests of the staggered halo loop issued from inside the MFMA burst, by position
# (python tools/build_ablate.py --conv3 "D:SMAP_STAG_DMA_IN_MFMA=1,SMAP_STAG_DMA_POS=3" ... POS=2 1 0 11 first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v18; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do for lib in smap_amd/libsmap_hip.so smap_amd/csrc/obj/libsmap_hip_conv3ablstag_dma_in_mfma1_stag_dma_pos*.so; do
  echo "lib $(basename $lib)" >> $O/ab.log
  [ $rep = 1 ] && SMAP_HIP_LIB=$PWD/$lib timeout 100 python tools/debug/halo_stag_race.py --runs 4 2>&1 | grep "RACE" >> $O/ab.log
  SMAP_HIP_LIB=$PWD/$lib timeout 100 python tools/bench_halo8.py --batch 16 --only 32,52,256,256 --tiles 45,44 2>&1 | grep -v amdgpu.ids >> $O/ab.log
  SMAP_HIP_LIB=$PWD/$lib timeout 100 python tools/bench_halo8.py --batch 16 --only 64,104,128,128 --tiles 44,45 2>&1 | grep -v amdgpu.ids >> $O/ab.log
done; done; cat $O/ab.log
}

v20() {
# visits 19-23: csrc/convc.hip (tile 94): probe, its tests, isolated timing (ablation / schedule variants: tools/build_ablate.py --one convc.hip
# SMAP_CONVC_ABLATE=N ..., selected with SMAP_HIP_LIB), in-situ A/B against the schedule without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v20; mkdir -p $O; export TMPDIR=/tmp
timeout 120 python tools/debug/convb_probe.py 94 2>&1 | grep -v amdgpu.ids > $O/probe.log; cat $O/probe.log
timeout 300 python -m pytest tests/test_backbone_gpu.py -q -x -m gpu -k "x94 or 128:94" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 120 python tools/bench_convb.py 94 91 --check 2>&1 | grep -v amdgpu.ids > $O/bench_convb.log; cat $O/bench_convb.log
bash tools/gpu_visits/ab_bench.sh $O/ab_block2.log 2 "--no-cpu-baseline --steps 40 --warmup 6" "SMAP_BLOCK=64:91" "" > /dev/null; cat $O/ab_block2.log
bash tools/gpu_visits/ab_bench.sh $O/ab_block2_d1.log 1 "--no-cpu-baseline --steps 30 --depth 1 --launch-frames 0" "SMAP_BLOCK=64:91" "" > /dev/null; cat $O/ab_block2_d1.log
}

v26() {
# visits 25, 26: 16 vs 32 frames per launch on the final schedule; the stem with 8- and 16-byte stores
# (python tools/build_ablate.py --one plan.hip SMAP_STEM_STORE16=0 first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v26; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do for lf in 16 32; do bash tools/gpu_visits/ab_bench.sh $O/ab_lf.log 1 "--no-cpu-baseline --steps 64 --warmup 8 --launch-frames $lf" "" > /dev/null; done; done; cat $O/ab_lf.log
for rep in 1 2; do for lib in smap_amd/libsmap_hip.so smap_amd/csrc/obj/libsmap_hip_plan_stem_store160.so; do
  SMAP_HIP_LIB=$PWD/$lib timeout 100 python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids >> $O/stem.log
done; done; cat $O/stem.log
}

# visits 5, 14, 24, 27: bash tools/gpu_visits/validate_all.sh <tag>
if [ -z "${1:-}" ] || ! declare -F "v$1" > /dev/null; then echo "usage: $0 <visit: 1 2 3 4 6 7 8 9 10 11 12 13 15 16 18 20 26>"; exit 2; fi
"v$1"
