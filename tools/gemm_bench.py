#!/usr/bin/env python3
"""Dev tool: time ag_gemm on the masked-forward GEMM shapes and check it against torch.matmul (checker only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops

dev = torch.device("cuda:0")
M = int(os.environ.get("GB_M", 100864))
only = os.environ.get("GB_ONLY")
shapes = [("qkv", 2304, 768, L.AG_EPI_BIAS), ("out", 768, 768, L.AG_EPI_BIAS_RESID), ("fc1", 3072, 768, L.AG_EPI_BIAS_GELU),
          ("fc2", 768, 3072, L.AG_EPI_BIAS_RESID)]
if only: shapes = [x for x in shapes if x[0] == only]
g = torch.Generator(device=dev); g.manual_seed(0)
for name, n, k, epi in shapes:
    a = (torch.rand((M, k), device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand((n, k), device=dev, generator=g) * 2 - 1) / k ** 0.5).to(torch.bfloat16)
    b = torch.rand(n, device=dev, generator=g)
    r = torch.rand((M, n), device=dev, generator=g).to(torch.bfloat16) if epi == L.AG_EPI_BIAS_RESID else None
    if os.environ.get("GB_ZERO") == "1":      # all-zero operands: the same instruction stream and cycles, less switching power (the clock the chip holds)
        a.zero_(); w.zero_()
        if r is not None: r.zero_()
    out = ops.gemm(a, w, b, epi, L.AG_BF16, resid=r)
    ref = a[:4096].float() @ w.float().T + b
    if epi == L.AG_EPI_BIAS_RESID: ref = ref + r[:4096].float()
    if epi == L.AG_EPI_BIAS_GELU: ref = torch.nn.functional.gelu(ref)
    err = (out[:4096].float() - ref).abs().max().item()
    tail = (out[-300:].float() - ((a[-300:].float() @ w.float().T + b) + (r[-300:].float() if r is not None else 0) if epi != L.AG_EPI_BIAS_GELU
            else torch.nn.functional.gelu(a[-300:].float() @ w.float().T + b))).abs().max().item()
    for _ in range(3): ops.gemm(a, w, b, epi, L.AG_BF16, resid=r, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): ops.gemm(a, w, b, epi, L.AG_BF16, resid=r, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:4s} M={M} N={n} K={k}: {ms*1e3:8.1f} us  {2.0*M*n*k/ms/1e9:7.1f} TF/s  max_err head={err:.3e} tail={tail:.3e}", flush=True)

# calibration only (NOT a product path): what the vendor library reaches on the same shapes
if os.environ.get("GB_TORCH", "0") == "1":
    for name, n, k, epi in shapes:
        a = (torch.rand((M, k), device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand((n, k), device=dev, generator=g) * 2 - 1) / k ** 0.5).to(torch.bfloat16)
        wt = w.t()
        for _ in range(3): torch.matmul(a, wt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): torch.matmul(a, wt)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"torch.matmul {name:4s} N={n} K={k}: {ms*1e3:8.1f} us  {2.0*M*n*k/ms/1e9:7.1f} TF/s (bf16 out, no epilogue)", flush=True)
