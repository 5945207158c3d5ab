#!/bin/bash
# round 4, visit 7: does the whole-Bottleneck kernel run two workgroups per CU?  occupancy query + 64 / 72 KiB LDS variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v7; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/bench_convb.py 91 90 93 92 > $O/convb_lds.log 2>&1
for n in lds64 lds72; do
  SMAP_HIP_LIB=$PWD/smap_amd/csrc/obj/libsmap_hip_convb$n.so timeout 200 python tools/bench_convb.py 91 90 93 92 >> $O/convb_lds.log 2>&1
done
grep "us per launch" $O/convb_lds.log
