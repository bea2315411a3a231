set -u
OUT=gpurun_out/r05g; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "ghost_select" > $OUT/t_ops.log 2>&1; tail -3 $OUT/t_ops.log
timeout 1500 python -m pytest tests/test_gpu_parallel.py -q -x -s > $OUT/t_par.log 2>&1; tail -8 $OUT/t_par.log
timeout 900 python tools/ghost_fraction.py 100 2 2 2 4 > $OUT/ghost_weak.json 2> $OUT/ghost_weak.log; tail -3 $OUT/ghost_weak.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "2 2 2"; do n=$(echo $g | tr -d " ")
  rocprofv3 --kernel-trace --stats -d $OUT/vprof_$n -o p -- python tools/ghost_fraction.py 100 $g 4 > /dev/null 2> $OUT/vranks_$n.log
  python tools/rocpd_stats.py $(ls $OUT/vprof_$n/*.db | head -1) $OUT/vranks_stats_$n.md > /dev/null; rm -rf $OUT/vprof_$n
  rocprofv3 --kernel-trace --stats -d $OUT/sprof_$n -o p -- python tools/ghost_fraction.py 100 $g 4 strong > /dev/null 2> $OUT/sranks_$n.log
  python tools/rocpd_stats.py $(ls $OUT/sprof_$n/*.db | head -1) $OUT/sranks_stats_$n.md > /dev/null; rm -rf $OUT/sprof_$n
done
tail -3 $OUT/vranks_222.log; head -30 $OUT/vranks_stats_222.md
