#!/usr/bin/env python3
"""Dev tool: the narrow-head (LTT side network: 12 heads x 8 dims) masked attention alone, ViT-base ladder shape R=1536 T=197 and the
packed BERT one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
R, T, H, heads = 1536, 197, 96, 12
qkv = torch.randn((R, T, 3 * H), device=dev, generator=g).to(torch.bfloat16)
bits = ops.pack_mask((torch.rand((R, T - 1), device=dev, generator=g) < 0.5).to(torch.int64))
fn = lambda: ops.masked_attention(qkv, bits, R, T, H, heads, 1, L.AG_MASK_VIT_MUL, L.AG_BF16)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 30 * 1e3
print(f"narrow vit R={R} T={T}: {us:.1f} us  ({(R*T*3*H*2 + R*T*H*2)/us/1e3:.0f} GB/s algorithmic)", flush=True)
