"""Diagnostic (variants/PTRACE.so only: cconv_pair.hip built with -DPX_TRACE, make -C dmcf_amd/csrc pair_trace): cycle stamps of
the phases of splat F's batch loop, summed over every 16th tile.
usage: cp variants/PTRACE.so dmcf_amd/libdmcf_hip.so; DMCF_CCONV_KERNEL=pair ONLY=L4 python tools/ptrace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import _lib
from tools import microbench

lib = ctypes.CDLL(os.path.join(ROOT, "dmcf_amd", "libdmcf_hip.so"))
buf = (ctypes.c_ulonglong * 24)()
microbench.main()
torch.cuda.synchronize()
lib.dmcf_ptrace(buf)
z = np.array(list(buf), dtype=np.float64)
names = ["prologue (first batch: idx -> pos -> geometry -> features)", "geometry + push_index", "issue: features, positions, indices",
         "splat of the first point's blocks (boundary batch)", "splat (+ merge of the first point)", "wait for the loads (fence)",
         "publish features", "records + classes", "(loop exit)", "merge of the second point + unpark", "", "", "", "(chunk loop exit)",
         "epilogue: sum over waves, bias, store", "", "chunk: tiles -> B rows", "chunk: barrier (slowest wave)",
         "chunk: contraction (B fragments, filter fragments, matrix)", "chunk: barrier", "partial sums -> LDS", "barrier"]
tot, waves, nb = z[10], z[11], z[12]
print(f"waves {waves:.0f} batches {nb:.0f} cycles/wave {tot / waves:.0f} cycles/batch {tot / nb:.0f}")
for k, n in enumerate(names):
    if not n: continue
    print(f"{n:60s} {100 * z[k] / tot:6.1f} %   {z[k] / nb:8.0f} cycles per batch")
