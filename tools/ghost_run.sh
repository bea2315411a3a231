# the sharded-rollout part of tools/profile_round.sh alone (ghost fractions, kernel time and dispatches per virtual rank)
set -u
OUT=${1:-gpurun_out/r05}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "ghost_select" > $OUT/t_ghost_ops.log 2>&1; tail -n 2 $OUT/t_ghost_ops.log
timeout 1500 python -m pytest tests/test_gpu_parallel.py -q -x -s > $OUT/t_ghost_par.log 2>&1; tail -n 3 $OUT/t_ghost_par.log
timeout 900 python tools/ghost_fraction.py 100 2 2 2 4 > $OUT/ghost_weak.json 2> $OUT/ghost_weak.log
timeout 900 python tools/ghost_fraction.py 100 2 2 2 4 strong > $OUT/ghost_strong.json 2> $OUT/ghost_strong.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "1 1 1" "2 1 1" "2 2 1" "2 2 2"; do n=$(echo $g | tr -d " ")
  rocprofv3 --kernel-trace --stats -d $OUT/vprof_$n -o p -- python tools/ghost_fraction.py 100 $g 4 > /dev/null 2> $OUT/vranks_$n.log
  python tools/rocpd_stats.py $(ls $OUT/vprof_$n/*.db | head -1) $OUT/vranks_stats_$n.md $OUT/vranks_calls_$n.txt > /dev/null; rm -rf $OUT/vprof_$n
done
for g in "2 1 1" "2 2 1" "2 2 2"; do n=$(echo $g | tr -d " ")
  rocprofv3 --kernel-trace --stats -d $OUT/sprof_$n -o p -- python tools/ghost_fraction.py 100 $g 4 strong > /dev/null 2> $OUT/sranks_$n.log
  python tools/rocpd_stats.py $(ls $OUT/sprof_$n/*.db | head -1) $OUT/sranks_stats_$n.md $OUT/sranks_calls_$n.txt > /dev/null; rm -rf $OUT/sprof_$n
done
for n in 111 211 221 222; do grep "^steps" $OUT/vranks_$n.log; tail -n 1 $OUT/vranks_stats_$n.md; done
for n in 211 221 222; do grep "^steps" $OUT/sranks_$n.log; tail -n 1 $OUT/sranks_stats_$n.md; done
