#!/bin/bash
# The round-end validation, one visit: the whole GPU suite, smoke(), the bench lines with timed-path parity (default, driver form, flip,
# RefineNet, batch-1 forward, f16), rocprofv3 --kernel-trace --stats of the driver's command, per-layer trace at depth 1, HBM traffic and
# MFMA utilisation from PMC passes (each its own run, with --kernel-trace only), host-budget rehearsal.
#     bash tools/gpu_visits/validate_all.sh <tag>      -> gpurun_out/<tag>/   (profiles/<round>_final_* are copies of these files)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
TAG=${1:-final}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py > $O/bench_x3.json 2> $O/bench_x3.err; echo "rc $?" >> $O/bench_x3.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_x3_driver_form.json 2>> $O/bench_x3.err
timeout 400 python bench.py --flip --steps 40 > $O/bench_x3_flip.json 2> $O/bench_x3_flip.err
timeout 400 python bench.py --refine --steps 60 > $O/bench_x3_refine.json 2> $O/bench_x3_refine.err
timeout 300 python bench.py --forward-only --batch 1 --graph --steps 300 --warmup 30 > $O/bench_x3_forward_b1.json 2>> $O/bench_x3.err
timeout 300 python bench.py --forward-only --batch 1 --steps 300 --warmup 30 > $O/bench_x3_forward_b1_kernel_by_kernel.json 2>> $O/bench_x3.err
timeout 300 python bench.py --precision f16 --steps 60 > $O/bench_f16.json 2>> $O/bench_x3.err
timeout 300 python bench.py --no-cpu-baseline --batch 1 --depth 1 --launch-frames 0 --steps 200 --warmup 20 > $O/bench_x3_batch1_full_path.json 2>> $O/bench_x3.err
TAG=$TAG python - <<'PY'
import json, os
for f in ("bench_x3", "bench_x3_driver_form", "bench_x3_flip", "bench_x3_refine", "bench_x3_forward_b1", "bench_x3_forward_b1_kernel_by_kernel", "bench_f16", "bench_x3_batch1_full_path"):
    try:
        d = json.load(open("gpurun_out/%s/%s.json" % (os.environ["TAG"], f))); c = d["config"]; m = c.get("e2e_parity") or {}
        print(f, round(d["value"], 1), "lf0", c.get("value_launch_frames_0"), "mfma frac", round(d["roofline"]["frac"], 4), "pipe", round(d["roofline"].get("pipe_frac", 0), 4),
              {k: m.get(k) for k in ("peaks_differing", "peaks_clear_mismatch", "max_joint_err_cm", "joints_over_0.1cm_unexplained", "lifter_tie_events", "timed_records_equal_these_frames")})
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp
# rocprofv3 --kernel-trace --stats of the driver's command
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_default -o smap -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/rocprof_default.log 2>&1
db=$(find $O/prof_default -name "*.db" | head -1); python $R/tools/prof_export.py $db $O/kernel_stats_default.csv; cp $(find $O/prof_default -name "*kernel_stats.csv" | head -1) $O/rocprofv3_stats_native_default.csv 2>/dev/null; rm -rf $O/prof_default
# per-layer, depth 1, one launch per step
SMAP_PRECISION=x3 SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_d1 -o smap -- python $R/bench.py --depth 1 --launch-frames 0 --steps 4 --warmup 2 --no-cpu-baseline > $O/rocprof_d1.log 2>&1
db=$(find $O/prof_d1 -name "*.db" | head -1); (cd $R; SMAP_PRECISION=x3 python tools/prof_layers.py $db 8 > $O/layers_d1.txt 2>&1); python $R/tools/prof_export.py $db $O/kernel_stats_d1.csv; rm -rf $O/prof_d1
# per-layer, depth 1, 16 frames per launch (what the default pipeline launches)
SMAP_PRECISION=x3 SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $O/prof_d1_16 -o smap -- python $R/bench.py --depth 1 --steps 6 --warmup 2 --no-cpu-baseline > $O/rocprof_d1_16.log 2>&1
db=$(find $O/prof_d1_16 -name "*.db" | head -1); (cd $R; SMAP_PRECISION=x3 python tools/prof_layers.py $db 16 > $O/layers_d1_16_frames.txt 2>&1); rm -rf $O/prof_d1_16
# HBM traffic: two separate PMC passes
SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pmc -- python $R/bench.py --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pmc -- python $R/bench.py --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
python $R/tools/prof_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/hbm_traffic_x3.json; cat $O/hbm_traffic_x3.json
rm -rf $O/pmc_fetch $O/pmc_write
# MFMA pipe utilisation, depth 1
SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o pmc -- python $R/bench.py --depth 1 --launch-frames 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
python $R/tools/prof_mfma.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/mfma_utilisation_x3.json > $O/mfma_utilisation_pmc.log 2>&1; cat $O/mfma_utilisation_pmc.log
rm -rf $O/pmc_mfma
# the same pass over the 16-frame launches the default pipeline issues
SMAP_BENCH_NO_LF0=1 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma16 -o pmc -- python $R/bench.py --depth 1 --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_mfma16.log 2>&1
python $R/tools/prof_mfma.py $(find $O/pmc_mfma16 -name "*counter_collection.csv" | head -1) $O/mfma_utilisation_x3_16_frames.json 16 > $O/mfma_utilisation_pmc_16_frames.log 2>&1; cat $O/mfma_utilisation_pmc_16_frames.log
rm -rf $O/pmc_mfma16
cd $R
bash tools/host_budget.sh 24 > $O/host_budget.log 2>&1; cat $O/host_budget.log
timeout 1500 python tools/cli_e2e.py --images 1024 --out $O/cli_e2e.json > $O/cli_e2e.log 2>&1; tail -7 $O/cli_e2e.log | cut -c1-400
