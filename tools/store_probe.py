#!/usr/bin/env python3
"""Dev tool: the output (store) path of a CU by store shape (ag_probe_store): the large-M GEMM writes a 128 KiB tile per workgroup."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L
dev = torch.device("cuda:0")
rows = 302592 // 256 * 256
for ncols in (768, 2304):
    buf = torch.empty((rows, ncols), dtype=torch.bfloat16, device=dev)
    for grid in (64, 256):
        for shape, name in ((0, "8 rows x 128 B"), (2, "4 rows x 256 B"), (1, "2 rows x 512 B")):
            for flags in (0, 1):
                b, g = C.c_double(), C.c_double()
                L.check(L.lib().ag_probe_store(shape, flags, buf.data_ptr(), ncols * 2, rows, grid, C.byref(b), C.byref(g), None))
                print(f"N={ncols} grid {grid:3d} {name} {'nt ' if flags else 'wb '}: {b.value:6.2f} B/clk/CU  {g.value/1e3:5.2f} TB/s chip   (a 128 KiB tile = {131072/b.value/1e3:5.1f} k cycles)")
