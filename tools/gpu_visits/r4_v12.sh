#!/bin/bash
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
