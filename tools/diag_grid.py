import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmcf_amd import ops, _lib
import ctypes
rng = np.random.default_rng(5)
p = rng.uniform(-2, 2, size=(20000, 3)).astype(np.float32)
far = np.concatenate([p[:500], p[:500] + np.float32([4000, 4000, 4000])])
pos = torch.from_numpy(far).cuda()
L = _lib.lib()
n = pos.shape[0]
vs = (ctypes.c_float * 3)(0.01, 0.01, 0.01)
nb = L.dmcf_grid_pos_workspace_bytes(n)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
rc = L.dmcf_grid_pos_bounds(pos.data_ptr(), n, vs, 0, None, 0, 0.1, ws.data_ptr(), nb, None)
torch.cuda.synchronize()
print(rc, ws[:24].view(torch.int32).tolist(), ws[24:40].view(torch.int64).tolist(), ws[40:52].view(torch.float32).tolist())
try:
    ops.grid_pos(pos, np.float32([0.01] * 3))
    print("no raise")
except Exception as e:
    print("raised", type(e), e)
