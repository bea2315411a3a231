#!/bin/bash
# Everything profiles/rNN_* is made from, in one gpurun call: bench lines, rocprofv3 kernel trace of the bench command,
# per-kernel HBM traffic (PMC, separate passes), SQ counters of every kernel above 3 % of the step, long rollouts.
# usage (on the GPU box): bash tools/profile_round.sh gpurun_out/<dir>;  then here: python tools/make_profiles.py gpurun_out/<dir> rNN
set -u
OUT=${1:-gpurun_out/profile_round}
mkdir -p $OUT
python bench.py --steps 5 --warmup 3 --layers-json $OUT/layers.json > $OUT/bench.log 2>&1
python bench.py --steps 20 --warmup 5 --cpu-side 0 > $OUT/bench_driver.log 2>&1
python bench.py --scene settled --steps 20 --warmup 5 --cpu-side 0 > $OUT/bench_settled.log 2>&1
for c in waterramps wbcsph; do python bench.py --config $c --steps 600 --warmup 20 > $OUT/bench_$c.log 2>&1; done
python bench.py --config liquid3d_dam --steps 20 --warmup 1 > $OUT/bench_liquid3d_dam_first20.log 2>&1
python bench.py --config liquid3d_dam --steps 200 --warmup 20 --cpu-side 0 > $OUT/bench_liquid3d_dam.log 2>&1
timeout 300 python tools/bench_scatter.py > $OUT/bench_scatter.log 2>&1
bash tools/sct_variants.sh > $OUT/sct_variants.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python bench.py --steps 5 --warmup 3 --cpu-side 0 > $OUT/prof.log 2>&1
python tools/rocpd_stats.py $(ls $OUT/prof/*.db | head -1) $OUT/kernel_stats.md > /dev/null
rm -rf $OUT/prof
bash tools/pmc_bench_traffic.sh $OUT/pmc_traffic > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc_traffic/FETCH_SIZE $OUT/pmc_traffic/WRITE_SIZE
# SQ counters, kernel by kernel on the micro-benchmark case that exercises it (default dispatch unless forced)
P=$OUT/pmc
bash tools/pmc_kernel.sh $P pair_L4 cconv_pair -- env DMCF_CCONV_KERNEL=pair ONLY=L4 python tools/microbench.py > $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P sct_S4 cconv_sct_kernel -- env ONLY=S4 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P pair_L3 cconv_pair -- env DMCF_CCONV_KERNEL=pair ONLY=L3 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P pair4_LQ cconv_pair -- env ONLY=LQ python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls1_L8 cconv_cls -- env ONLY=L8 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls2_L5 cconv_cls -- env ONLY=L5 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls2n_L6 cconv_cls -- env ONLY=L6 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P cls4_LP cconv_cls -- env ONLY=LP python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P z3_L14 cconv_z3 -- env DMCF_CCONV_KERNEL=z3 ONLY=L14 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P ws_L14 cconv_ws -- env ONLY=L14 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P ws_L2 cconv_ws -- env ONLY=L2 python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P direct_ASCC cconv_direct -- env ONLY=ASCC python tools/microbench.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P frs frs_query_padded -- python tools/bench_search.py >> $OUT/pmc.txt 2>&1
bash tools/pmc_kernel.sh $P lat lat_conv_kernel -- python tools/bench_lattice.py >> $OUT/pmc.txt 2>&1
python tools/bench_search.py > $OUT/search.log 2>&1
python tools/bench_lattice.py > $OUT/lattice.log 2>&1
timeout 600 python tools/microbench.py > $OUT/microbench.log 2>&1
# the short-row layers under every kernel that serves them (round 5: wave specialisation), and the producers' / consumers' stamps
for k in ws z3 cls pair; do echo "== DMCF_CCONV_KERNEL=$k" >> $OUT/microbench_short.log; DMCF_CCONV_KERNEL=$k ONLY=L14,L2,L5,IN timeout 300 python tools/microbench.py 2>&1 | grep pairs >> $OUT/microbench_short.log; done
cp dmcf_amd/libdmcf_hip.so /tmp/product.so; cp variants/WTRACE.so dmcf_amd/libdmcf_hip.so
for L in L14 L2 L5 IN; do DMCF_CCONV_KERNEL=ws ONLY=$L timeout 200 python tools/wtrace.py 2>&1 | grep -v amdgpu.ids >> $OUT/wtrace.log; done
cp /tmp/product.so dmcf_amd/libdmcf_hip.so
# (round 5 committed four tracebacks here: a stale variants/WTRACE.so.  The variants are rebuilt before every profile round --
# `make -C dmcf_amd/csrc ws_trace sct_variants` in the build container -- and a traceback in the stamps fails the script loudly)
if grep -q "Traceback" $OUT/wtrace.log; then echo "WTRACE FAILED: rebuild variants/ (make -C dmcf_amd/csrc ws_trace)"; fi
# the small configurations: host profile + synchronisations, and a kernel trace of 100 steady steps each
for r in waterramps wbcsph liquid3d_dam; do timeout 300 python tools/profile_small.py $r 100 > $OUT/small_$r.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/sprof_$r -o p -- python tools/profile_small.py $r 100 > /dev/null 2>&1
  python tools/rocpd_stats.py $(ls $OUT/sprof_$r/*.db | head -1) $OUT/small_kernel_stats_$r.md $OUT/small_by_calls_$r.txt > /dev/null; rm -rf $OUT/sprof_$r
done
for r in "liquid3d_dam 200" "waterramps 600" "wbcsph 3200"; do set -- $r; timeout 900 python tools/long_rollout.py $1 $2 --out $OUT/rollout_$1.json >> $OUT/rollouts.log 2>&1; done
timeout 900 python tools/ghost_fraction.py 100 2 2 2 4 > $OUT/ghost_weak.json 2> $OUT/ghost_weak.log
timeout 900 python tools/ghost_fraction.py 100 2 2 2 4 strong > $OUT/ghost_strong.json 2> $OUT/ghost_strong.log
# (the virtual-rank kernel traces of rounds 4 / 5 are not repeated: profiles/r05_virtual_rank_kernel_time.md)
tail -1 $OUT/bench.log | cut -c1-300; tail -1 $OUT/bench_driver.log | cut -c1-200
