#!/bin/bash
# Everything profiles/rNN_* is made from, in one gpurun call: bench lines, rocprofv3 kernel trace of the bench command,
# per-kernel HBM traffic (PMC, separate passes), SQ counters of every kernel above 3 % of the step, long rollouts.
# usage (on the GPU box): bash tools/profile_round.sh gpurun_out/<dir>;  then here: python tools/make_profiles.py gpurun_out/<dir> rNN
set -u
OUT=${1:-gpurun_out/profile_round}
mkdir -p $OUT
python bench.py --steps 5 --warmup 3 --layers-json $OUT/layers.json > $OUT/bench.log 2>&1
python bench.py --steps 20 --warmup 5 --cpu-side 0 > $OUT/bench_driver.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python bench.py --steps 5 --warmup 3 --cpu-side 0 > $OUT/prof.log 2>&1
python tools/rocpd_stats.py $(ls $OUT/prof/*.db | head -1) $OUT/kernel_stats.md > /dev/null
rm -rf $OUT/prof
bash tools/pmc_bench_traffic.sh $OUT/pmc_traffic > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc_traffic/FETCH_SIZE $OUT/pmc_traffic/WRITE_SIZE
# SQ counters, kernel by kernel on the micro-benchmark case that exercises it (default dispatch unless forced)
P=$OUT/pmc
bash tools/pmc_kernel.sh $P pair_L4 cconv_pair -- env DMCF_CCONV_KERNEL=pair ONLY=L4 python tools/microbench.py > $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P pair_L3 cconv_pair -- env DMCF_CCONV_KERNEL=pair ONLY=L3 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P pair4_LQ cconv_pair -- env ONLY=LQ python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls1_L8 cconv_cls -- env ONLY=L8 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls2_L5 cconv_cls -- env ONLY=L5 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls2n_L6 cconv_cls -- env ONLY=L6 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls4_LP cconv_cls -- env ONLY=LP python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P z3_L14 cconv_z3 -- env ONLY=L14 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P direct_ASCC cconv_direct -- env ONLY=ASCC python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P frs frs_query_padded -- python tools/bench_search.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P lat lat_conv_kernel -- python tools/bench_lattice.py >> $OUT/pmc.txt 2>&1
python tools/bench_search.py > $OUT/search.log 2>&1
python tools/bench_lattice.py > $OUT/lattice.log 2>&1
timeout 600 python tools/microbench.py > $OUT/microbench.log 2>&1
for r in "liquid3d_dam 200" "waterramps 600" "wbcsph 3200"; do set -- $r; timeout 900 python tools/long_rollout.py $1 $2 --out $OUT/rollout_$1.json >> $OUT/rollouts.log 2>&1; done
timeout 900 python tools/ghost_fraction.py 100 2 2 2 4 > $OUT/ghost_weak.json 2> $OUT/ghost_weak.log
timeout 900 python tools/ghost_fraction.py 50 2 2 2 4 > $OUT/ghost_strong.json 2> $OUT/ghost_strong.log
# kernel time per virtual rank (rocprofv3 kernel trace of 4 steps) for 1 / 2 / 4 / 8 ranks: the input of DESIGN section 6's scaling model
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "1 1 1" "2 1 1" "2 2 1" "2 2 2"; do n=$(echo $g | tr -d " ")
  rocprofv3 --kernel-trace --stats -d $OUT/vprof_$n -o p -- python tools/ghost_fraction.py 100 $g 4 > /dev/null 2> $OUT/vranks_$n.log
  python tools/rocpd_stats.py $(ls $OUT/vprof_$n/*.db | head -1) $OUT/vranks_stats_$n.md > /dev/null; rm -rf $OUT/vprof_$n
done
tail -1 $OUT/bench.log | cut -c1-300; tail -1 $OUT/bench_driver.log | cut -c1-200
