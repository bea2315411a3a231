set -u
OUT=gpurun_out/pmc_lat
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "lat_conv" --output-format csv -d $OUT/$name -o p -- python tools/bench_lattice.py > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC
python tools/pmc_parse2.py $OUT sq1 sq2 sq3
