#!/bin/bash
# GPU visit 2: full GPU test-suite (incl. reference-build comparison + CLI e2e), tile-variant A/B, profile.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
cat /sys/fs/cgroup/cpu.max > $O/cgroup.txt 2>&1; nproc >> $O/cgroup.txt
echo "== smoke" ; timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -2 $O/smoke.log
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -40 $O/pytest_gpu.log
run_bench() { # name, remap
  SMAP_TILE_REMAP="$2" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$1.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1.log").read().strip().splitlines()[-1])
    print("$1", "fps=%.1f step=%.2fms backbone=%.2fms TF=%.0f" % (d["value"], d["ms_per_step"], d["roofline"]["backbone_ms_per_batch"], d["roofline"]["achieved"]))
except Exception as e:
    print("$1 FAILED", e); print(open("$O/bench_$1.log").read()[-1500:])
PY
}
echo "== bench A/B"
run_bench base ""
run_bench deep_all "0:5,1:6,2:7,3:8,4:9"
run_bench deep_128 "0:5"
run_bench deep_64 "2:7"
run_bench deep_12864 "1:6"
echo "== bench with cpu baseline" ; timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1 ; tail -1 $O/bench.log | cut -c1-1500
echo "== rocprof deep_all" ; cd /tmp && export TMPDIR=/tmp
SMAP_TILE_REMAP="0:5,1:6,2:7,3:8,4:9" timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_deep -o smap -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_deep.log 2>&1 ; echo "rocprof rc=$?"
