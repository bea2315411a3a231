"""Diagnostic (variants/WTRACE.so only: cconv_ws.hip built with -DWS_TRACE, make -C dmcf_amd/csrc ws_trace): cycle stamps of
the producers' and the consumers' phases in the wave-specialised CConv kernel, summed over every 16th workgroup.
usage: cp variants/WTRACE.so dmcf_amd/libdmcf_hip.so; DMCF_CCONV_KERNEL=ws ONLY=L14 python tools/wtrace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import _lib
from tools import microbench

lib = ctypes.CDLL(os.path.join(ROOT, "dmcf_amd", "libdmcf_hip.so"))
buf = (ctypes.c_ulonglong * 32)()
lib.dmcf_wtrace(buf)
microbench.main()
torch.cuda.synchronize()
lib.dmcf_wtrace(buf)
z = np.array(list(buf), dtype=np.float64)
tot, tiles, nb = z[8], z[9], z[10]
print(f"PRODUCER: tiles {tiles:.0f} batches {nb:.0f} ({nb / tiles:.2f} per tile)  clocks per tile {tot / tiles:.0f}")
names = ["stages before the splat: geometry, index push, feature / position / index loads", "splat segments", "merge + row store + clear (per point)",
         "wait at 'full'", "ring refill + tile context", "wait at 'free' (the consumers' pull)", "publish features, records, classes", "(pin)"]
for k, n in enumerate(names):
    print(f"  {n:85s} {100 * z[k] / tot:6.1f} %   {z[k] / tiles:8.0f} clocks per tile")
ctot = z[20]
print(f"CONSUMER: clocks per tile {ctot / tiles:.0f}")
for k, n in enumerate(["wait at 'full' (the producers)", "pull + previous tile's sums", "wait at 'free'", "contraction + partial sums"]):
    print(f"  {n:85s} {100 * z[16 + k] / ctot:6.1f} %   {z[16 + k] / tiles:8.0f} clocks per tile")
