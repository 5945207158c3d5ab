#!/bin/bash
# round 4, visit 10: in-situ coordinate descent over the tiles of the heaviest 16-frame shapes on the new schedule (whole-block launches on)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v10; mkdir -p $O
export TMPDIR=/tmp SMAP_BENCH_NO_LF0=1
timeout 840 python tools/insitu_tune.py --precision x3 --steps 40 --warmup 6 --candidates tools/insitu_candidates_r4.json --out $O/tile_table_x3_insitu.json > $O/insitu.log 2>&1
cat $O/insitu.log
