#!/bin/bash
# Round 2, GPU visit 4: in-schedule flip-TTA (parity), pipeline tests with lazy records, in-situ A/B: depth 3, BK=32-only table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_entry_gpu.py -m gpu -q -p no:cacheprovider -k "flip or pipeline or cli or headsum or module_forward or full_size" 2>&1 | tail -8 | tee $O/r2_04_tests.log
for i in 1 2; do
for v in "default:" "bk32:SMAP_TILE_TABLE_X3=$R/tools/tables/x3_bk32_only.json"; do
  name=${v%%:*}; envs=${v#*:}
  for d in 2 3; do
    echo "-- x3 table=$name depth $d" | tee -a $O/r2_04_ab.log
    env $envs timeout 300 python bench.py --depth $d --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_04_ab.log
  done
done
done
echo "-- f16 (lazy records)" | tee -a $O/r2_04_ab.log
timeout 300 python bench.py --precision f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_04_ab.log
echo "-- x3 do_flip pipeline fps" | tee -a $O/r2_04_ab.log
timeout 300 python bench.py --flip --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_04_ab.log
timeout 300 python bench.py --flip --precision f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-230 | tee -a $O/r2_04_ab.log
