#!/usr/bin/env python3
"""Dev tool: timing of the mid-size GEMM kernel (gemm.hip) on the shapes of the explainer training step and of single-input
forwards.  Env: AG_GEMM_NST / AG_GEMM_BT (kernel knobs), AG_GEMM_BIG_MIN_TILES (ring-kernel threshold)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
dev = torch.device("cuda:0")
import itertools
if os.environ.get("GS_FWD"):   # single-input .. 8-input forward shapes of ViT-base (B x 32 masks x 197 tokens)
    SHAPES = [(b * 6304, n, k) for b in (1, 2, 3, 4, 6, 8, 16) for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072))
              if not os.environ.get("GS_N768") or n == 768]
else:
  SHAPES = [(1576, 768, 768), (1576, 2304, 768), (1576, 3072, 768), (1576, 768, 3072), (768, 768, 1600), (3072, 768, 1600),
          (768, 3072, 1600), (1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072), (768, 768, 1024), (6304, 768, 768), (6304, 768, 3072)]
EPI = int(os.environ.get("GS_EPI", L.AG_EPI_BIAS_F32))
res = []
for m, n, k in SHAPES:
    a = (torch.rand((m, k), device=dev) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand((n, k), device=dev) * 2 - 1) / k ** 0.5).to(torch.bfloat16)
    b = torch.rand(n, device=dev)
    out = torch.empty((m, n), dtype=torch.float32 if EPI == L.AG_EPI_BIAS_F32 else torch.bfloat16, device=dev)
    res_t = (torch.rand((m, n), device=dev) * 2 - 1).to(torch.bfloat16) if EPI == L.AG_EPI_BIAS_RESID else None
    st = ops.new_row_stats(m, n, dev) if os.environ.get("GS_STATS") and res_t is not None else None
    for _ in range(10): ops.gemm(a, w, b, EPI, L.AG_BF16, out=out, resid=res_t, stats_out=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.gemm(a, w, b, EPI, L.AG_BF16, out=out, resid=res_t, stats_out=st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    ref = a.float() @ w.float().t() + b + (res_t.float() if res_t is not None else 0)
    err = float((out.float() - ref).abs().max())
    res.append(f"{m}x{n}x{k}: {us:5.1f}us {2.0*m*n*k/us/1e6:5.0f}TF err {err:.1e}")
print(f"NST={os.environ.get('AG_GEMM_NST','-')} BT={os.environ.get('AG_GEMM_BT','-')} MIN_TILES={os.environ.get('AG_GEMM_BIG_MIN_TILES','-')}\n  " + "\n  ".join(res), flush=True)
