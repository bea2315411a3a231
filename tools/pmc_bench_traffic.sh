#!/bin/bash
# HBM traffic of the CConv kernels over the bench workload: FETCH_SIZE and WRITE_SIZE in separate passes
# (kernel-trace + pmc only), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
OUT=${1:-gpurun_out/pmc_bench}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "cconv|lat_conv" --output-format csv -d $OUT/$c -o p -- \
      python bench.py --steps 2 --warmup 1 --cpu-side 0 > $OUT/$c.log 2>&1
done
python - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
tot = {}
n = 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(f"{out}/{c}/p_counter_collection.csv")) if r["Counter_Name"] == c and "pack_filter" not in r["Kernel_Name"] and "lat_build" not in r["Kernel_Name"]]
    tot[c] = sum(float(r["Counter_Value"]) for r in rows)
    n = len(rows)
# units: KB (rocprofv3 derived metric); gfx950 correction: FETCH_SIZE counts 64 B per 128 B request on wide streams -> x2
res = dict(launches=n, fetch_kb_raw=tot["FETCH_SIZE"], write_kb_raw=tot["WRITE_SIZE"],
           hbm_bytes_per_launch=(2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / max(n, 1),
           hbm_bytes_per_launch_uncorrected=(tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / max(n, 1),
           note="sum over all dmcf::cconv* dispatches of `bench.py --steps 2 --warmup 1` (3 steps x 17 launches: 13 neighbour-list layers + 4 lattice layers), rocprofv3 --pmc, "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide reads); WRITE_SIZE as reported")
json.dump(res, open(f"{out}/cconv_hbm_traffic.json", "w"), indent=1)
print(res)
PY
