#!/usr/bin/env python3
"""Dev tool: what the residual epilogue costs the K=768 / K=3072 N=768 GEMMs (same shapes with and without the residual read)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
if os.environ.get("GB_LIB"): L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["GB_LIB"])
dev = torch.device("cuda:0"); M = int(os.environ.get("GB_M", 302592))
def mk(n, k): return ((torch.rand((n, k), device=dev) * 2 - 1) / k ** 0.5).to(torch.bfloat16)
out = torch.empty((M, 768), dtype=torch.bfloat16, device=dev)
r = (torch.rand((M, 768), device=dev) * 2 - 1).to(torch.bfloat16)
bias = torch.rand(768, device=dev)
for k in (768, 3072):
    a = (torch.rand((M, k), device=dev) * 2 - 1).to(torch.bfloat16); w = mk(768, k)
    st = torch.zeros((M, 2), dtype=torch.float32, device=dev)
    for name, epi, rr, so in (("bias", L.AG_EPI_BIAS, None, None), ("bias+resid", L.AG_EPI_BIAS_RESID, r, None), ("bias+resid+stats", L.AG_EPI_BIAS_RESID, r, st)):
        for _ in range(10): ops.gemm(a, w, bias, epi, L.AG_BF16, resid=rr, out=out, stats_out=so)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.gemm(a, w, bias, epi, L.AG_BF16, resid=rr, out=out, stats_out=so)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f"K={k} N=768 {name:17s}: {us:7.1f} us  {2.0 * M * 768 * k / us / 1e6:6.0f} TF", flush=True)
