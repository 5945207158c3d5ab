#!/bin/bash
# round 4, visit 11: eight-wave halo 3x3 tiles (40..43): parity of the new instances, then cold isolated timings against the shipped tiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v11; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_backbone_gpu.py -q -x -m gpu -k "x1x40 or x1x41 or x1x42 or x1x43 or halo_conv" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 240 python tools/bench_halo8.py --batch 16 > $O/halo8_b16.log 2>&1; cat $O/halo8_b16.log
timeout 200 python tools/bench_halo8.py --batch 8 > $O/halo8_b8.log 2>&1; cat $O/halo8_b8.log
