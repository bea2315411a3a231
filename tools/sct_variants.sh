#!/bin/bash
# Diagnostic: splat S with one part removed each (make -C dmcf_amd/csrc sct_variants), timed by tools/bench_scatter.py; the
# product library is put back at the end.
set -e
cp dmcf_amd/libdmcf_hip.so /tmp/libdmcf_hip.keep
for f in variants/SCT_*.so; do
  cp $f dmcf_amd/libdmcf_hip.so
  echo "== $f"
  timeout 300 python tools/bench_scatter.py 2>&1 | grep "scatter\|rror" || true
done
cp /tmp/libdmcf_hip.keep dmcf_amd/libdmcf_hip.so
