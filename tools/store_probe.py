#!/usr/bin/env python3
"""Dev tool: what the CU's store path makes of a 256 x 256 bf16 output tile by store shape (ag_probe_store): 0 = the shipped epilogue
(8 rows x 128 B per store after an LDS transposition), 1 / 2 wider rows, 3 / 4 = straight from the 16x16 MFMA accumulator layout
(16 rows x 4 x 16 B at a 32-byte stride / 16 rows x 64 B).  Prints B/clk/CU and GB/s, plain and non-temporal."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L
dev = torch.device("cuda:0")
rows, ld = 302592 // 256 * 256, 3072 * 2
buf = torch.empty((rows, ld), dtype=torch.uint8, device=dev)
for grid in [int(a) for a in sys.argv[1:]] or [256]:
    for shape in (0, 1, 2, 3, 4):
        for flags in (0, 1):
            bpc, gbs = C.c_double(), C.c_double()
            L.check(L.lib().ag_probe_store(shape, flags, buf.data_ptr(), ld, rows, grid, C.byref(bpc), C.byref(gbs), None))
            print(f"grid {grid} shape {shape} nt {flags}: {bpc.value:6.1f} B/clk/CU  {gbs.value:7.0f} GB/s  ({131072 / bpc.value / 2.0e3:5.2f} us per tile at 2 GHz)", flush=True)
