"""Masked forward at a list of (workload, inputs, masks) points: fwd/s eager and graph-replayed, per-kernel-class time per step from
the in-library hipEvents.  `python tools/fwd_points.py vit_base:1 vit_base:4 vit_large:1:64 vit_large:1:8` (masks 0 = the config's K).
Round-5 development tool for the under-filled-launch path (small batches, 8-GPU strong-scaling shards)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from autognothi_amd import _lib as L, engine  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    engine.set_precision("bf16")
    pts = sys.argv[1:] or ["vit_base:1", "vit_base:4"]
    tag = os.environ.get("FP_TAG", "")
    for pt in pts:
        f = pt.split(":")
        wl, b, k = f[0], int(f[1]), int(f[2]) if len(f) > 2 else 0
        job = bench.Job(wl, dev, 0, 1, b, k, "bf16")
        n = int(os.environ.get("FP_STEPS", "30"))
        for _ in range(3):
            job.step()
        torch.cuda.synchronize()
        L.check(L.lib().ag_profile_enable(1))
        for c in bench.EPI_NAMES:
            bench.collect(c)
        bench.collect(10)
        el, _ = bench.timed(job.step, n, 2, None, dev)
        L.check(L.lib().ag_profile_enable(0))
        st = {bench.EPI_NAMES[c]: bench.collect(c) for c in bench.EPI_NAMES}
        try:
            st["gemm_ex"] = bench.collect(10)
        except Exception:
            pass
        gstep = engine.GraphedStep(job.step)
        el_g, _ = bench.timed(gstep, n, 2, None, dev)
        del gstep
        f_exec = bench.flops_executed(job.kind, job.params, job.T, job.K, 1.0)
        tot = sum(v[0] for v in st.values())
        # (the profiler's events cover the settle loop too: per-step figures are totals scaled to the timed steps' share)
        nsteps_prof = max(1.0, tot / (1e3 * el / n)) if tot > 0 else 1.0
        out = {"tag": tag, "point": pt, "rows": job.R, "eager_fwd_s": round(job.R * n / el, 1), "graph_fwd_s": round(job.R * n / el_g, 1),
               "eager_ms": round(1e3 * el / n, 4), "graph_ms": round(1e3 * el_g / n, 4),
               "exec_frac_eager": round(job.R * n / el * f_exec / 1e12 / bench.PEAK_BF16_TFLOPS, 4),
               "exec_frac_graph": round(job.R * n / el_g * f_exec / 1e12 / bench.PEAK_BF16_TFLOPS, 4),
               "classes": {k_: {"avg_us": round(1e3 * v[0] / max(1, v[3]), 2), "n": v[3], "tflops": round(v[1] / max(v[0], 1e-9) / 1e9, 1)}
                           for k_, v in st.items() if v[3]}}
        print(json.dumps(out), flush=True)
        del job


if __name__ == "__main__":
    main()
