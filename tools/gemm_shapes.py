#!/usr/bin/env python3
"""Dev tool: sustained timing of the four encoder GEMM shapes (ViT-base, R*T = 100864 rows) through ag_gemm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
if os.environ.get("GB_LIB"):   # A/B: an alternative build of the library (same box, same run)
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["GB_LIB"])
dev = torch.device("cuda:0"); M = int(os.environ.get("GB_M", 100864))
ZERO = os.environ.get("GB_ZERO") == "1"
def mk(n, k): return ((torch.rand((n, k), device=dev) * 2 - 1) / k ** 0.5).to(torch.bfloat16) * (0 if ZERO else 1)
x768 = (torch.rand((M, 768), device=dev) * 2 - 1).to(torch.bfloat16) * (0 if ZERO else 1)
x3072 = (torch.rand((M, 3072), device=dev) * 2 - 1).to(torch.bfloat16) * (0 if ZERO else 1)
r = (torch.rand((M, 768), device=dev) * 2 - 1).to(torch.bfloat16) * (0 if ZERO else 1)
shapes = [("qkv", x768, mk(2304, 768), L.AG_EPI_BIAS, None), ("proj", x768, mk(768, 768), L.AG_EPI_BIAS_RESID, r),
          ("fc1", x768, mk(3072, 768), L.AG_EPI_BIAS_GELU, None), ("fc2", x3072, mk(768, 3072), L.AG_EPI_BIAS_RESID, r)]
outs = {n: torch.empty((M, w.shape[0]), dtype=torch.bfloat16, device=dev) for n, _, w, _, _ in shapes}
bias = {n: torch.rand(w.shape[0], device=dev) for n, _, w, _, _ in shapes}
def run(n, a, w, e, rr): ops.gemm(a, w, bias[n], e, L.AG_BF16, resid=rr, out=outs[n])
for _ in range(20):
    for sh in shapes: run(*sh)
torch.cuda.synchronize()
res = []
tot = 0.0
for sh in shapes:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): run(*sh)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3; tot += us
    fl = 2.0 * M * sh[2].shape[0] * sh[2].shape[1]
    res.append(f"{sh[0]} {us:6.1f}us {fl/us/1e6:6.0f}TF")
print(" | ".join(res), f"| sum {tot:7.1f}us", flush=True)
