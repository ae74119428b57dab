#!/usr/bin/env python3
"""Dev tool: what the GELU epilogue costs the fc1 GEMM (M = 302 592, N = 3072, K = 768): bias vs bias+GELU, with / without LN fold."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
if os.environ.get("GB_LIB"): L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["GB_LIB"])
dev = torch.device("cuda:0"); M = int(os.environ.get("GB_M", 302592))
a = (torch.rand((M, 768), device=dev) * 2 - 1).to(torch.bfloat16)
w = ((torch.rand((3072, 768), device=dev) * 2 - 1) / 768 ** 0.5).to(torch.bfloat16)
bias = torch.rand(3072, device=dev); out = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev)
for name, epi in (("bias", L.AG_EPI_BIAS), ("bias+gelu", L.AG_EPI_BIAS_GELU), ("bias", L.AG_EPI_BIAS), ("bias+gelu", L.AG_EPI_BIAS_GELU)):
    for _ in range(10): ops.gemm(a, w, bias, epi, L.AG_BF16, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.gemm(a, w, bias, epi, L.AG_BF16, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"fc1 {name:10s}: {us:7.1f} us  {2.0 * M * 3072 * 768 / us / 1e6:6.0f} TF", flush=True)
