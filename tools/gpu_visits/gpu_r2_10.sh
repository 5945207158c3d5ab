#!/bin/bash
# Round 2, GPU visit 10: single-stage / half-epilogue variants (tiles 60..62): parity spot check + cold timings vs tiles 20/50
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python - <<'PY' 2>&1 | tail -12 | tee $O/r2_10_parity.log
import sys
sys.path[:0] = ['.', 'tests', 'tests/golden']
import torch
from test_backbone_gpu import _run_single_conv
for x3 in (True, False):
    for case in [(3, 10, 14, 192, 320, 3, 1, 60, True, True, True), (2, 16, 24, 256, 256, 1, 1, 61, True, True, False), (2, 16, 24, 64, 256, 1, 1, 62, False, True, False), (1, 32, 52, 256, 512, 1, 2, 60, False, False, False)]:
        got, ref, cout = _run_single_conv(*case, seed=7, x3=x3)
        err = (got[..., :cout] - ref).abs().max().item() / ref.abs().max().item()
        print('x3' if x3 else 'f16', case[:8], 'rel err %.2e' % err, 'OK' if err < (3e-6 if x3 else 2e-3) else 'FAIL')
PY
for t in 20 50 60 61 62; do
  echo "== x3 tile $t cold" | tee -a $O/r2_10_stream.log
  python tools/bench_conv.py --x3 --iters 30 --rotate 3 --only L3,L1,L7 --tile-override L3:$t,L1:$t,L7:$t 2>/dev/null | tee -a $O/r2_10_stream.log
done
for t in 4 50 60 62; do
  echo "== f16 tile $t cold" | tee -a $O/r2_10_stream.log
  python tools/bench_conv.py --iters 30 --rotate 5 --only L3,L1,L7 --tile-override L3:$t,L1:$t,L7:$t 2>/dev/null | tee -a $O/r2_10_stream.log
done
