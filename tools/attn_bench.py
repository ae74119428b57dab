#!/usr/bin/env python3
"""Dev tool: sustained timing of the masked-attention kernel on the bench shapes (ViT-base R=1536 T=197, BERT T=512
fixed-length and packed), through ag_masked_attention / ag_masked_attention_varlen.  GB_LIB=<file in lib/> A/Bs a build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
if os.environ.get("GB_LIB"):
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["GB_LIB"])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
only = os.environ.get("ATTN_ONLY")
for name, R, T, H, heads, mode in (("vit_base", 1536, 197, 768, 12, L.AG_MASK_VIT_MUL), ("vit_1head", 18432, 197, 64, 1, L.AG_MASK_VIT_MUL),
                                   ("bert_fixed", 512, 512, 768, 12, L.AG_MASK_BERT_ADD)):
    if only and name != only:
        continue
    qkv = torch.randn((R, T, 3 * H), device=dev, generator=g).to(torch.bfloat16)
    keep = torch.rand((R, T - 1), device=dev, generator=g) < 0.5
    bits = ops.pack_mask(keep.to(torch.int64))
    us = timeit(lambda: ops.masked_attention(qkv, bits, R, T, H, heads, 1, mode, L.AG_BF16))
    fl = 4.0 * R * T * T * H
    out.append(f"{name} {us:7.1f}us {fl / us / 1e6:5.0f}TF")
print(" | ".join(out), flush=True)
