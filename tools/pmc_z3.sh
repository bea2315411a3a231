#!/bin/bash
# PMC passes for one forced CConv kernel on one microbench case.  usage: tools/pmc_z3.sh <outdir> <kernel> <case>
set -u
OUT=${1:-gpurun_out/pmc_z3}; K=${2:-z3}; L=${3:-L3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; DMCF_CCONV_KERNEL=$K ONLY=$L timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "cconv" --output-format csv -d $OUT/$name -o p -- python tools/microbench.py > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
python - "$OUT" <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
for name in ("sq1", "sq2", "sq3"):
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in agg.items():
            print(name, k, {c: f"{v:.4g}" for c, v in d.items()})
PY
