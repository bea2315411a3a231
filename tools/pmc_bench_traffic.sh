#!/bin/bash
# HBM traffic of the CConv kernels over the bench workload: FETCH_SIZE and WRITE_SIZE in separate passes
# (kernel-trace + pmc only), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Writes <out>/cconv_hbm_traffic.json with
# the bytes per launch of every kernel instantiation (``by_kernel``; bench.py cites the dominant kernel's entry).
set -u
OUT=${1:-gpurun_out/pmc_bench}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "cconv|lat_conv|frs_query|frs_fix" --output-format csv -d $OUT/$c -o p -- \
      python bench.py --steps 2 --warmup 1 --cpu-side 0 > $OUT/$c.log 2>&1
done
python - "$OUT" <<'PY'
import csv, json, re, sys, collections
out = sys.argv[1]
per = collections.defaultdict(lambda: dict(FETCH_SIZE=0.0, WRITE_SIZE=0.0, n=0))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{c}/p_counter_collection.csv")):
        if r["Counter_Name"] != c or "pack_filter" in r["Kernel_Name"] or "lat_build" in r["Kernel_Name"] or "pack_direct" in r["Kernel_Name"]:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        per[name][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            per[name]["n"] += 1
# units: KB (rocprofv3 derived metric); gfx950 correction: FETCH_SIZE counts 64 B per 128 B request on wide streams -> x2
by = {k: dict(launches=v["n"], fetch_kb_raw=v["FETCH_SIZE"], write_kb_raw=v["WRITE_SIZE"],
              hbm_bytes_per_launch=(2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / max(v["n"], 1)) for k, v in per.items()}
n = sum(v["n"] for v in per.values())
res = dict(launches=n, by_kernel=by,
           hbm_bytes_per_launch=sum(v["hbm_bytes_per_launch"] * v["launches"] for v in by.values()) / max(n, 1),
           note="dmcf::cconv* / lat_conv dispatches of `bench.py --steps 2 --warmup 1` (3 steps x 17 launches: 13 neighbour-list layers + 4 "
                "lattice layers), rocprofv3 --kernel-trace --pmc, FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 reports 64 B per 128-B request); WRITE_SIZE as reported")
json.dump(res, open(f"{out}/cconv_hbm_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in by.items()}))
PY
