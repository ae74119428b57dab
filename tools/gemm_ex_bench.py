"""Timing of ag_gemm_ex on the training step's Linear shapes (forward NT / dX NN / dW TN) over split counts, next to the round-3
path for the same product (ag_gemm on pre-cast / pre-transposed operands, cast + transpose launches not counted) and the vendor
library (torch.matmul, calibration only).  One JSON line per shape; bench.py's secondary.under_filled_gemm uses `sweep()`."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops  # noqa: E402


def _time(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def shapes(rows, h=768, i=3072):
    """(label, order, M, N, Kc) of one transformer block's Linear GEMMs on `rows` token rows."""
    out = []
    for name, n, k in (("qkv", 3 * h, h), ("o", h, h), ("fc1", i, h), ("fc2", h, i)):
        out.append((f"{name}.fwd", ops.NT, rows, n, k))
        out.append((f"{name}.dX", ops.NN, rows, k, n))
        out.append((f"{name}.dW", ops.TN, n, k, rows))
    return out


def sweep(rows, dev, splits_list=(1, 2, 3, 4, 6, 8), vendor=True, old=True):
    res = []
    for label, order, m, n, kc in shapes(rows):
        a = torch.randn((kc, m) if order[0] else (m, kc), device=dev).to(torch.bfloat16)
        b = torch.randn((kc, n) if order[1] else (n, kc), device=dev).to(torch.bfloat16)
        fl = 2.0 * m * n * kc
        rec = ops.gemm_ex_splits(m, n, kc)
        row = {"shape": label, "M": m, "N": n, "Kc": kc, "recommended_splits": rec, "us": {}}
        for s in splits_list:
            if s > max(1, ((kc + 63) // 64) // 2):
                continue
            if s == 1:
                row["us"]["store_s1"] = round(_time(lambda: ops.gemm_ex(a, b, order, L.AG_EX_STORE, out_dtype=ops.BF16)), 2)
            slabs = torch.empty((s, m, n), dtype=torch.float32, device=dev)
            row["us"][f"slabs_s{s}"] = round(_time(lambda: ops.gemm_ex(a, b, order, L.AG_EX_SLABS, splits=s, out=slabs)), 2)
        best = min(row["us"].values())
        row["best_tflops"] = round(fl / best / 1e6, 1)
        if old:   # the round-3 route: NT kernel of gemm.hip on operands already in NT form
            a_nt = (a.t().contiguous() if order[0] else a)
            b_nt = (b.t().contiguous() if order[1] else b)
            if a_nt.shape[1] % 64 == 0:
                row["us"]["r3_ag_gemm_nt"] = round(_time(lambda: ops.gemm(a_nt, b_nt, None, L.AG_EPI_BIAS_F32, L.AG_BF16, m=m)), 2)
        if vendor:
            a_v = a.t() if order[0] else a
            b_v = b if order[1] else b.t()
            row["us"]["vendor_matmul"] = round(_time(lambda: torch.matmul(a_v, b_v)), 2)
        res.append(row)
    return res


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for rows in [int(x) for x in (sys.argv[1:] or ["1576", "1024", "6304"])]:
        for r in sweep(rows, dev):
            r["rows"] = rows
            print(json.dumps(r), flush=True)
