#!/bin/bash
# Diagnostic: time micro-benchmark cases with each library variant under variants/ (built by hand); the last one stays installed.
for f in variants/*.so; do
  cp $f dmcf_amd/libdmcf_hip.so
  echo "== $f"
  timeout 600 python tools/microbench.py 2>&1 | grep "^L\|^ASCC\|^IN\|rror"
done
