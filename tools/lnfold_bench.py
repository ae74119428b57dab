import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops
dev = torch.device("cuda:0"); M, N, K = 100864, 2304, 768
a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
w = ((torch.rand((N, K), device=dev) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
b = torch.rand(N, device=dev); s = w.float().sum(1).contiguous(); st = ops.row_stats(a)
out = ops.gemm(a, w, b, L.AG_EPI_BIAS, L.AG_BF16)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
print("plain   %.1f us" % t(lambda: ops.gemm(a, w, b, L.AG_EPI_BIAS, L.AG_BF16, out=out)))
print("folded  %.1f us" % t(lambda: ops.gemm(a, w, b, L.AG_EPI_BIAS, L.AG_BF16, out=out, ln_stats=st, ln_colsum=s, ln_eps=1e-12)))
r = torch.rand((M, 768), device=dev).to(torch.bfloat16); w2 = w[:768].contiguous(); o2 = torch.empty((M, 768), dtype=torch.bfloat16, device=dev)
so = torch.zeros((M, 2), device=dev)
print("resid   %.1f us" % t(lambda: ops.gemm(a, w2, None, L.AG_EPI_BIAS_RESID, L.AG_BF16, resid=r, out=o2)))
print("resid+stats %.1f us" % t(lambda: ops.gemm(a, w2, None, L.AG_EPI_BIAS_RESID, L.AG_BF16, resid=r, out=o2, stats_out=so)))
