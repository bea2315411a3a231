#!/bin/bash
# Diagnostic: time the L8 micro-benchmark case with each library variant under variants/ (built by hand).
for f in variants/*.so; do
  cp $f dmcf_amd/libdmcf_hip.so
  echo "== $f"
  DMCF_CCONV_KERNEL=${KERNEL:-blk} timeout 300 python tools/microbench.py 2>&1 | grep "^L8\|^L6\|rror"
done
