OUT=${1:-gpurun_out/r05s}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in waterramps; do
  rocprofv3 --kernel-trace --stats -d $OUT/sprof_$r -o p -- python tools/profile_small.py $r 100 > /dev/null 2>&1
  python tools/rocpd_stats.py $(ls $OUT/sprof_$r/*.db | head -1) $OUT/small_kernel_stats_$r.md $OUT/small_calls_$r.txt > /dev/null; rm -rf $OUT/sprof_$r
done
