OUT=${1:-gpurun_out/r05s}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mfma or matrix_core or cconv" 2>&1 | tail -n 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 2 3; do for r in wbcsph; do
  DMCF_MFMA_DEBUG=$d timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/sprof_$r -o p -- python tools/profile_small.py $r 60 > /dev/null 2>&1
  python tools/rocpd_stats.py $(ls $OUT/sprof_$r/*.db | head -1) $OUT/dbg${d}_$r.md > /dev/null; rm -rf $OUT/sprof_$r
  echo "dbg $d"; grep "cconv_mfma_kernel<false, false, 1, 4>" $OUT/dbg${d}_$r.md | cut -c1-140
done; done
