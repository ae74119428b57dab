#!/usr/bin/env python3
"""Dev tool: LayerNorm kernel timing on the encoder and ladder row shapes (bf16 in/out)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
dev = torch.device("cuda:0")
for rows, h in ((302592, 96), (302592, 768)):
    x = torch.randn((rows, h), device=dev).to(torch.bfloat16); g = torch.rand(h, device=dev); b = torch.rand(h, device=dev)
    for _ in range(5): ops.layernorm(x, g, b, 1e-12, L.AG_BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.layernorm(x, g, b, 1e-12, L.AG_BF16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"rows {rows} H {h}: {us:7.1f} us  {rows * h * 4 / us / 1e6:6.2f} TB/s", flush=True)
