#!/bin/bash
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
