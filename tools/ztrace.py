"""Diagnostic (variants/TRACE.so only: cconv_z3.hip built with -DZX_TRACE): cycle stamps of the
phases of splat E's batch loop, summed over every 16th tile.  usage: cp variants/TRACE.so dmcf_amd/libdmcf_hip.so; ONLY=L3 python tools/ztrace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import _lib
from tools import microbench

lib = ctypes.CDLL(os.path.join(ROOT, "dmcf_amd", "libdmcf_hip.so"))
buf = (ctypes.c_ulonglong * 16)()
microbench.main()
torch.cuda.synchronize()
lib.dmcf_ztrace(buf)
z = np.array(list(buf), dtype=np.float64)
names = ["prologue", "publish0+idx+issue1", "splat0", "geom+order+push_index", "publish1+pos+issue0", "splat1", "push_rec",
         "last batch", "(unused)", "epilogue", "", "", "", "merge + first barrier (imbalance)", "contraction", "reduction barrier"]
tot, waves, nb = z[10], z[11], z[12]
print(f"waves {waves:.0f} batches {nb:.0f} cycles/wave {tot / waves:.0f} cycles/batch {tot / nb:.0f}")
for k, n in enumerate(names):
    if not n: continue
    print(f"{n:28s} {100 * z[k] / tot:6.1f} %   {z[k] / nb:8.0f} cycles per batch")
