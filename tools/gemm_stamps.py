#!/usr/bin/env python3
"""Dev tool: run the diagnostic (stamped) build of the ring GEMM once and print where a half-step's cycles go."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AG_GEMM_DBG"] = "/tmp/ag_dbg_ptr.txt"
from autognothi_amd import _lib as L, ops
if os.environ.get("GB_LIB"):
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["GB_LIB"])
dev = torch.device("cuda:0")
M, N, K = 100864, 2304, int(os.environ.get("GB_K", 768))
a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
w = ((torch.rand((N, K), device=dev) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
b = torch.rand(N, device=dev)
for _ in range(3):
    out = ops.gemm(a, w, b, L.AG_EPI_BIAS, L.AG_BF16)
torch.cuda.synchronize()
ptr = int(open("/tmp/ag_dbg_ptr.txt").read().strip(), 16)
n = 2 * 8 * 128 * 8
host = (ctypes.c_ulonglong * n)()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(host, ctypes.c_void_p(ptr), n * 8, 2)
d = np.frombuffer(host, dtype=np.uint64).reshape(2, 8, 128, 8).astype(np.int64)
nh = K // 32
names = ["barA", "read+dma", "lgkm+vm", "barB", "mfma", "loop"]
for blk in (0, 1):
    for wave in (0, 3, 4, 7):
        s = d[blk, wave]
        seg = np.zeros(6)
        lo, hi = 3, nh - 4
        for j in range(lo, hi):
            for k in range(5):
                seg[k] += s[j, k + 1] - s[j, k]
            seg[5] += s[j + 1, 0] - s[j, 5]
        cnt = hi - lo
        print(f"blk{blk} wave{wave}: " + " ".join(f"{n} {v/cnt:5.0f}" for n, v in zip(names, seg)) + f" | total/iter {(s[hi,0]-s[lo,0])/cnt:6.0f}")

for blk in (0, 1):
    for wave in (0, 4):
        s = d[blk, wave]
        t0 = s[120, 0]
        print(f"blk{blk} wave{wave}: entry 0 | prologue issued {s[121,0]-t0} | slot0 landed {s[122,0]-t0} | loop end {s[123,0]-t0} "
              f"| epilogue issued {s[124,0]-t0} | stores acked {s[125,0]-t0}")
