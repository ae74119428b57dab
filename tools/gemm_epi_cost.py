#!/usr/bin/env python3
"""Dev tool: what each epilogue of the large-M GEMM costs on the N = 768 shapes (out-proj K = 768, fc2 K = 3072): bias only / + residual /
+ residual + row statistics / fp32 output, and the per-tile fixed cost from the two K values (linear fit)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
if os.environ.get("GB_LIB"): L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["GB_LIB"])
dev = torch.device("cuda:0"); M = int(os.environ.get("GB_M", 302592)); N = int(os.environ.get("GB_N", 768))
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {}
for K in (768, 3072):
    a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand((N, K), device=dev) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    b = torch.rand(N, device=dev); r = torch.rand((M, N), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev); st = ops.new_row_stats(M, N, dev)
    res[K] = dict(bias=t(lambda: ops.gemm(a, w, b, L.AG_EPI_BIAS, L.AG_BF16, out=out)),
                  gelu=t(lambda: ops.gemm(a, w, b, L.AG_EPI_BIAS_GELU, L.AG_BF16, out=out)),
                  resid=t(lambda: ops.gemm(a, w, b, L.AG_EPI_BIAS_RESID, L.AG_BF16, resid=r, out=out)),
                  resid_stats=t(lambda: ops.gemm(a, w, b, L.AG_EPI_BIAS_RESID, L.AG_BF16, resid=r, out=out, stats_out=st)))
    del a, w, r, out
tiles = ((M + 255) // 256) * ((N + 255) // 256) / 256.0
for k in res[768]:
    per_step = (res[3072][k] - res[768][k]) / tiles / 36.0
    fixed = res[768][k] / tiles - 12 * per_step
    print(f"{k:12s} K=768 {res[768][k]:7.1f} us  K=3072 {res[3072][k]:7.1f} us  -> per K=64 step {per_step:5.2f} us, per-tile fixed {fixed:5.1f} us ({tiles:.2f} tile rounds)")
