#!/usr/bin/env python3
"""Dev tool: the global -> LDS feed rate of a CU (ag_probe_dma) over wave counts, barrier on / off, L2-resident vs HBM stream,
LDS-DMA vs plain loads.  The ring GEMM needs 32 KiB per half-step and CU."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L
dev = torch.device("cuda:0")
K = int(os.environ.get("GB_K", 3072)); ld = K * 2; nh = K // 32
panels = 1182
A = (torch.rand((panels * 256, K), device=dev) * 2 - 1).to(torch.bfloat16)
W = (torch.rand((768, K), device=dev) * 2 - 1).to(torch.bfloat16)
for waves in [int(x) for x in os.environ.get('GB_WAVES', '4,8,16').split(',')]:
    for flags in [int(x) for x in os.environ.get('GB_FLAGS', '0,1,2,3,8,9,4,6,12').split(',')]:
        b, g, z = C.c_double(), C.c_double(), C.c_double()
        L.check(L.lib().ag_probe_dma(waves, nh, flags, A.data_ptr(), W.data_ptr(), ld, panels, C.byref(b), C.byref(g), C.byref(z), None))
        what = ("barrier " if flags & 1 else "free    ") + ("L2-hot " if flags & 2 else ("L2-own " if flags & 8 else "stream ")) + ("plain" if flags & 4 else "ldsdma") + (" 8x128B" if flags & 16 else " 16x64B")
        print(f"waves {waves:2d} {what}: {b.value:6.2f} B/clk/CU  {g.value/1e3:6.2f} TB/s chip  {z.value:.2f} GHz   (a 32 KiB half-step = {32768/b.value:6.0f} cycles)")
