#!/bin/bash
# SQ counter passes for ONE kernel: three separate rocprofv3 runs (--kernel-trace --pmc only, as MI355X_MICROARCH.md prescribes),
# summed over the launches of the command.
#   tools/pmc_kernel.sh <outdir> <label> <kernel-regex> -- <command ...>
# e.g.  tools/pmc_kernel.sh gpurun_out/pmc pair_L4 cconv_pair -- env DMCF_CCONV_KERNEL=pair ONLY=L4 python tools/microbench.py
set -u
OUT=$1; LABEL=$2; REGEX=$3; shift 4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$REGEX" --output-format csv -d $OUT/${LABEL}_$name -o p -- "${CMD[@]}" > $OUT/${LABEL}_$name.log 2>&1; }
CMD=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
# the clock of the launches themselves (the busy fractions are over SIMD-CYCLES: assuming 2.4 GHz gave 110 % for the search, r04)
run sq4 GRBM_GUI_ACTIVE
python - "$OUT" "$LABEL" <<'PY'
import csv, sys, collections, glob, json
out, label = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for name in ("sq1", "sq2", "sq3", "sq4"):
    for f in glob.glob(f"{out}/{label}_{name}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "pack" in k or "lat_build" in k:
                continue
            k = k.split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if name == "sq1" and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                launches[k] += 1
res = {k: dict(launches=launches[k], **v) for k, v in agg.items()}
json.dump(res, open(f"{out}/{label}.json", "w"), indent=1)
for k, v in res.items():
    print(label, k, {c: f"{x:.4g}" if isinstance(x, float) else x for c, x in v.items()})
PY
rm -rf $OUT/${LABEL}_sq1 $OUT/${LABEL}_sq2 $OUT/${LABEL}_sq3 $OUT/${LABEL}_sq4
