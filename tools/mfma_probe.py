#!/usr/bin/env python3
"""Dev tool: power-capped MFMA ceiling of this board (ag_probe_mfma), random vs zero operands."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autognothi_amd import _lib as L
torch.cuda.init(); torch.zeros(1, device="cuda")
for zero in (0, 2, 0, 2, 1, 3):
    for iters in (20000, 200000):
        tf, ghz = C.c_double(), C.c_double()
        L.check(L.lib().ag_probe_mfma(iters, zero, C.byref(tf), C.byref(ghz), None))
        print(f"{'32x32x16' if zero & 2 else '16x16x32'} operands {'zero' if zero & 1 else 'random'} iters {iters}: {tf.value:7.1f} TFLOP/s at {ghz.value:.3f} GHz effective "
              f"({100 * tf.value / (ghz.value * 1024 * 1.024) if ghz.value else 0:.1f} % of 1024 flop/clk/SIMD x 1024 SIMDs at that clock)")
