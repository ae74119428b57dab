#!/usr/bin/env python3
"""Dev tool: fused side MLP (ag_side_mlp) vs the three launches it replaces, LTT ladder shape (h=96, I=384), M rows."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
dev = torch.device("cuda:0")
BF = L.AG_BF16
for m in (100864, 302592):
    h, i = 96, 384
    x = torch.randn(m, h, device=dev).to(torch.bfloat16)
    w1 = (torch.randn(i, h, device=dev) / h ** 0.5).to(torch.bfloat16); b1 = torch.randn(i, device=dev)
    w2 = (torch.randn(h, i, device=dev) / i ** 0.5).to(torch.bfloat16); b2 = torch.randn(h, device=dev)
    g, be = torch.ones(h, device=dev), torch.zeros(h, device=dev)
    def fused(): return ops.side_mlp(x, w1, b1, w2, b2, g, be, 1e-12, False)
    def three():
        u, _ = ops.layernorm(x, g, be, 1e-12, BF)
        f = ops.gemm(u, w1, b1, L.AG_EPI_BIAS_GELU, BF)
        return ops.gemm(f, w2, b2, L.AG_EPI_BIAS_RESID, BF, resid=x)
    for name, fn in (("fused", fused), ("3 launches", three)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
        print(f"M={m} {name:11s} {dt * 1e6:8.1f} us   {4.0 * m * h * i / dt / 1e12:6.1f} TFLOP/s   {m * h * 4 / dt / 1e9:7.1f} GB/s (x + out)")
    print("max |fused - 3 launches| =", float((fused().float() - three().float()).abs().max()))
