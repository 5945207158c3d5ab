cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SMAP_TAIL="64:80" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tail -o t -- python $R/bench.py --depth 1 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
ST=$(ls $R/gpurun_out/prof_tail/*kernel_stats.csv $R/gpurun_out/prof_tail/*/*kernel_stats.csv 2>/dev/null | head -1)
head -14 $ST | cut -c1-200
grep -i "tail" $ST | cut -c1-250
rm -rf $R/gpurun_out/prof_tail
