#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R/_r2 && python -m smap_amd.build > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
for d in _r2 .; do
  tag=$(echo $d | tr -d './_'); tag=${tag:-now}
  (cd $R/$d && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1_$tag -o b1 -- python bench.py --forward-only --batch 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/b1_$tag.log 2>&1)
  ST=$(ls $O/b1_$tag/*kernel_stats.csv $O/b1_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $d"; tail -1 $O/b1_$tag.log | cut -c1-200
  python - "$ST" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels", len(rows), "total ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:90]:90} calls {r["Calls"]:>5} total {float(r["TotalDurationNs"])/1e6:8.3f} ms avg {float(r["AverageNs"])/1e3:7.1f} us')
PY
  rm -rf $O/b1_$tag
done
