#!/usr/bin/env python3
"""Dev tool (round 6): interleaved A/B of knob settings on the four encoder GEMMs of the benchmarked step, in ONE process.

  R6_VARIANTS="name:K=V,K=V;name2:..."   (default: phased vs pipelined main loop; the key LIB=<suffix> selects a variant build)
  R6_M (302592)  R6_ROUNDS (7)  R6_REPS (12)  R6_ONLY=qkv|out|fc1|fc2  R6_SHAPES="name:N:K:epilogue;..." (other models' Linears)

Every round runs every variant once per shape (order rotated per round); a variant's environment is applied and the library's
knobs re-read (ag_reload_knobs) before its launches.  Reports median and min of the per-launch time (hipEvents around REPS
launches) and TFLOP/s, plus bit-equality of every variant's output with the first variant's."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops

dev = torch.device("cuda:0")
M = int(os.environ.get("R6_M", 302592))
ROUNDS = int(os.environ.get("R6_ROUNDS", 7))
REPS = int(os.environ.get("R6_REPS", 12))
only = os.environ.get("R6_ONLY")
spec = os.environ.get("R6_VARIANTS", "phased:AG_GEMM_PIPE=0;pipe:AG_GEMM_PIPE=1")
variants = []
for part in spec.split(";"):
    name, _, kv = part.partition(":")
    env = dict(x.split("=", 1) for x in kv.split(",") if x)
    variants.append((name, env))
all_keys = sorted({k for _, e in variants for k in e if k not in ("LIB", "DATA")})
LIBDIR = os.path.dirname(L.LIB_PATH)


def apply(env):
    # LIB=<suffix>: lib/libautognothi_hip_<suffix>.so (tools/build_variant.sh) instead of the shipped library
    L.use_library(os.path.join(LIBDIR, f"libautognothi_hip_{env['LIB']}.so") if env.get("LIB") else None)
    for k in all_keys:
        if k in env: os.environ[k] = env[k]
        else: os.environ.pop(k, None)
    ops.reload_knobs()


shapes = [("qkv", 2304, 768, L.AG_EPI_BIAS), ("out", 768, 768, L.AG_EPI_BIAS_RESID), ("fc1", 3072, 768, L.AG_EPI_BIAS_GELU),
          ("fc2", 768, 3072, L.AG_EPI_BIAS_RESID)]
if os.environ.get("R6_SHAPES"):      # "name:N:K:epilogue;..." (epilogue 0 bias / 1 bias+gelu / 2 bias+residual)
    shapes = [(f[0], int(f[1]), int(f[2]), int(f[3])) for f in (x.split(":") for x in os.environ["R6_SHAPES"].split(";"))]
if only: shapes = [x for x in shapes if x[0] in only.split(",")]
g = torch.Generator(device=dev); g.manual_seed(0)
for name, n, k, epi in shapes:
    a = (torch.rand((M, k), device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand((n, k), device=dev, generator=g) * 2 - 1) / k ** 0.5).to(torch.bfloat16)
    b = torch.rand(n, device=dev, generator=g)
    r = torch.rand((M, n), device=dev, generator=g).to(torch.bfloat16) if epi == L.AG_EPI_BIAS_RESID else None
    outs, times = {}, {v[0]: [] for v in variants}
    # DATA=zero: the same launches on all-zero operands (same instruction stream and cycles; what differs is switching power, i.e. the clock
    # the chip holds: MI355X_MICROARCH.md 'DVFS give-back' item 1)
    az, wz = torch.zeros_like(a), torch.zeros_like(w)
    pick = lambda env: (az, wz) if env.get("DATA") == "zero" else (a, w)   # noqa: E731
    for vn, env in variants:      # warm-up + outputs
        apply(env)
        a_, w_ = pick(env)
        o = ops.gemm(a_, w_, b, epi, L.AG_BF16, resid=r)
        for _ in range(3): ops.gemm(a_, w_, b, epi, L.AG_BF16, resid=r, out=o)
        torch.cuda.synchronize()
        outs[vn] = o.clone()
    ref = a[:2048].float() @ w.float().T + b
    if epi == L.AG_EPI_BIAS_RESID: ref = ref + r[:2048].float()
    if epi == L.AG_EPI_BIAS_GELU: ref = torch.nn.functional.gelu(ref)
    first = variants[0][0]
    o = torch.empty_like(outs[first])
    for rd in range(ROUNDS):
        order = variants[rd % len(variants):] + variants[:rd % len(variants)]
        for vn, env in order:
            apply(env)
            a_, w_ = pick(env)
            ops.gemm(a_, w_, b, epi, L.AG_BF16, resid=r, out=o)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS): ops.gemm(a_, w_, b, epi, L.AG_BF16, resid=r, out=o)
            e1.record(); torch.cuda.synchronize()
            times[vn].append(e0.elapsed_time(e1) / REPS * 1e3)
    for vn, _ in variants:
        t = times[vn]
        med, mn = statistics.median(t), min(t)
        err = (outs[vn][:2048].float() - ref).abs().max().item()
        same = bool(torch.equal(outs[vn], outs[first]))
        print(f"{name:4s} M={M} N={n} K={k} {vn:10s}: median {med:8.1f} us ({2.0*M*n*k/med/1e6:7.1f} TF/s)  min {mn:8.1f} us ({2.0*M*n*k/mn/1e6:7.1f})"
              f"  max_err {err:.3e}  bit_equal_to_{first}={same}", flush=True)
    del a, w, b, r, outs, o
    torch.cuda.empty_cache()
