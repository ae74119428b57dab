#!/usr/bin/env python3
"""Dev tool: the ring GEMM on large square problems (the shapes the guide's 8-phase template is quoted on), uniform [-1,1) operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
dev = torch.device("cuda:0")
for n in (4096, 8192):
    for k in (768, n):
        a = (torch.rand((n, k), device=dev) * 2 - 1).to(torch.bfloat16)
        w = (torch.rand((n, k), device=dev) * 2 - 1).to(torch.bfloat16)
        b = torch.zeros(n, device=dev)
        out = torch.empty((n, n), dtype=torch.bfloat16, device=dev)
        for _ in range(5): ops.gemm(a, w, b, L.AG_EPI_BIAS, L.AG_BF16, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, w, b, L.AG_EPI_BIAS, L.AG_BF16, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"M=N={n} K={k}: {us:8.1f} us  {2.0 * n * n * k / us / 1e6:6.0f} TF", flush=True)
        ref = torch.matmul(a, w.t())
        for _ in range(3): torch.matmul(a, w.t(), out=ref)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): torch.matmul(a, w.t(), out=ref)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"   torch.matmul (hipBLASLt) calibration: {us:8.1f} us  {2.0 * n * n * k / us / 1e6:6.0f} TF", flush=True)
