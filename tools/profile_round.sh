#!/bin/bash
# Everything profiles/rNN_* is made from, in one gpurun call: bench lines, rocprofv3 kernel trace of the bench command,
# per-kernel HBM traffic (PMC, separate passes), PMC counters of the dominant kernel, long rollouts.
# usage (on the GPU box): bash tools/profile_round.sh gpurun_out/<dir>
set -u
OUT=${1:-gpurun_out/profile_round}
mkdir -p $OUT
python bench.py --steps 5 --warmup 3 --layers-json $OUT/layers.json > $OUT/bench.log 2>&1
python bench.py --steps 20 --warmup 5 --cpu-side 0 > $OUT/bench_driver.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python bench.py --steps 5 --warmup 3 --cpu-side 0 > $OUT/prof.log 2>&1
python tools/rocpd_stats.py $(ls $OUT/prof/*.db | head -1) $OUT/kernel_stats.md > /dev/null
bash tools/pmc_bench_traffic.sh $OUT/pmc_traffic > $OUT/pmc_traffic.log 2>&1
bash tools/pmc_z3.sh $OUT/pmc_z3 z3 L3 > $OUT/pmc_z3.txt 2>&1
for r in "liquid3d_dam 200" "waterramps 600" "wbcsph 3200"; do set -- $r; timeout 900 python tools/long_rollout.py $1 $2 --out $OUT/rollout_$1.json >> $OUT/rollouts.log 2>&1; done
rm -rf $OUT/prof $OUT/pmc_traffic/FETCH_SIZE $OUT/pmc_traffic/WRITE_SIZE $OUT/pmc_z3/sq1 $OUT/pmc_z3/sq2 $OUT/pmc_z3/sq3
tail -1 $OUT/bench.log | cut -c1-300; tail -1 $OUT/bench_driver.log | cut -c1-200
