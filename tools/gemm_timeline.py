#!/usr/bin/env python3
"""Dev tool: per-workgroup timeline of the whole-line GEMM (AG_GEMM_DBG stamps: start, epilogue issued, stores acknowledged, hardware id):
workgroup life, store-acknowledge wait, and the gap between a workgroup's end and its successor's start on the same CU."""
import ctypes, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AG_GEMM_DBG"] = "/tmp/ag_dbg_ptr.txt"
from autognothi_amd import _lib as L, ops
if os.environ.get("GB_LIB"): L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["GB_LIB"])
dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("GB_M", 302592)), int(os.environ.get("GB_N", 768)), int(os.environ.get("GB_K", 768))
epi = {"resid": L.AG_EPI_BIAS_RESID, "gelu": L.AG_EPI_BIAS_GELU, "bias": L.AG_EPI_BIAS}[os.environ.get("GB_EPI", "resid")]
fold = os.environ.get("GB_FOLD", "0") == "1"      # LayerNorm-folded consumer (QKV, fc1 of the pre-LN encoder)
a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
w = ((torch.rand((N, K), device=dev) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
b = torch.rand(N, device=dev); r = torch.rand((M, N), device=dev).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
kw = {}
if fold: kw = dict(ln_stats=ops.row_stats(a), ln_colsum=w.float().sum(1).contiguous(), ln_eps=1e-6)
if epi == L.AG_EPI_BIAS_RESID: kw = dict(resid=r, stats_out=ops.new_row_stats(M, N, dev))
for _ in range(20):
    ops.gemm(a, w, b, epi, L.AG_BF16, out=out, **kw)
torch.cuda.synchronize()
ptr = int(open("/tmp/ag_dbg_ptr.txt").read().strip(), 16)
nwg = ((M + 255) // 256) * ((N + 255) // 256)
n = 8 * nwg
host = (ctypes.c_ulonglong * n)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(host, ctypes.c_void_p(ptr), n * 8, 2)
d = np.frombuffer(host, dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
t0, t1, t2, hw, tp, tl = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4], d[:, 5]
print(f"prologue (start -> first step) median {np.median(tp - t0) * 0.01:.2f} us | main loop {np.median(tl - tp) * 0.01:.2f} us | epilogue (loop end -> issued) {np.median(t1 - tl) * 0.01:.2f} us")
life, ack = (t1 - t0) * 0.01, (t2 - t1) * 0.01
print(f"{nwg} workgroups; life (start -> epilogue issued) median {np.median(life):.2f} us; stores acknowledged after a further median {np.median(ack):.2f} us (p10 {np.percentile(ack,10):.2f}, p90 {np.percentile(ack,90):.2f})")
slots = collections.defaultdict(list)
for i in range(nwg):
    slots[(i & 7, int(hw[i]) & 0x7F00)].append((t0[i], t2[i]))
gaps = []
for k, v in slots.items():
    v.sort()
    gaps += [(v[j + 1][0] - v[j][1]) * 0.01 for j in range(len(v) - 1)]
gaps = np.asarray(gaps)
print(f"{len(slots)} CU slots; gap from 'stores acknowledged' to the successor's start: median {np.median(gaps):.2f} us (p10 {np.percentile(gaps,10):.2f}, p90 {np.percentile(gaps,90):.2f})")
print(f"kernel span {(t2.max() - t0.min()) * 0.01:.1f} us; spread of first-round starts {np.ptp(np.sort(t0)[:256]) * 0.01:.2f} us")
