"""Per-route timing of ag_gemm_ws on the masked forward's Linears at under-filled sizes (calibration of ag_ws_plan's cost model).
`python tools/ws_bench.py 6304:768 1576:1024 ...` (rows:hidden; intermediate = 4 x hidden).  Back-to-back launches, L2-warm."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops  # noqa: E402


def _time(fn, iters=30, warm=4):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    for spec in sys.argv[1:] or ["6304:768", "1576:1024"]:
        m, h = [int(x) for x in spec.split(":")]
        i = 4 * h
        for name, n, k, epi in (("qkv", 3 * h, h, L.AG_EPI_BIAS), ("o", h, h, L.AG_EPI_BIAS_RESID), ("fc1", i, h, L.AG_EPI_BIAS_GELU),
                                ("fc2", h, i, L.AG_EPI_BIAS_RESID)):
            a = torch.randn((m, k), device=dev).to(torch.bfloat16)
            w = (torch.randn((n, k), device=dev) / k ** 0.5).to(torch.bfloat16)
            b = torch.randn(n, device=dev)
            r = torch.randn((m, n), device=dev).to(torch.bfloat16) if epi == L.AG_EPI_BIAS_RESID else None
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            st = torch.empty(((n + 127) // 128, m, 2), dtype=torch.float32, device=dev) if r is not None else None
            row = {"rows": m, "shape": name, "N": n, "K": k, "us": {}}
            routes = [("gemm", 0, 0), ("ex", 2, 0)]
            if r is not None:
                routes += [("big_split", 1, 0)] + [(f"slabs{s}", 3, s) for s in (1, 2, 3, 4, 6, 8)]
            for tag, rt, sp in routes:
                try:
                    fn = lambda: ops.gemm_ws(a, w, b, epi, resid=r, out=out, stats_out=st, out_cols_ok=2 if rt == 2 else 1, route=rt, splits=sp)  # noqa: E731
                    fn()
                    row["us"][tag] = round(_time(fn), 2)
                except RuntimeError:
                    pass
            _, cols = ops.gemm_ws(a, w, b, epi, resid=r, out=out, stats_out=st, out_cols_ok=3)
            row["planned_cols"] = cols
            row["us"]["planned"] = round(_time(lambda: ops.gemm_ws(a, w, b, epi, resid=r, out=out, stats_out=st, out_cols_ok=3)), 2)
            fl = 2.0 * m * n * k
            row["best_tflops"] = round(fl / min(row["us"].values()) / 1e6, 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
