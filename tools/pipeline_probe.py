"""Dev probe (round 4): one ViT-base encoder layer (LN-folded QKV, masked attention, out-proj + residual + stats, fc1 + GELU, fc2 + residual + stats)
on R x 197 token rows, twelve times: (a) the shipped order on one stream; (b) the batch as two halves on two CU-partitioned streams — the GEMMs of one
half on c CUs of every XCD while the attention of the other half runs on the remaining 32 - c (events between the streams).  ms per layer."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autognothi_amd import _lib as L, ops  # noqa: E402
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
R, T, H, I, heads, LAYERS = int(os.environ.get("PP_R", "1536")), 197, 768, 3072, 12, 12
BF = L.AG_BF16
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(s, device=dev, generator=g) * 0.5).to(torch.bfloat16)  # noqa: E731
w_qkv, w_o, w_fc1, w_fc2 = rnd(3 * H, H) * 0.07, rnd(H, H) * 0.07, rnd(I, H) * 0.07, rnd(H, I) * 0.04
b_qkv, b_o, b_fc1, b_fc2 = (torch.zeros(n, device=dev) for n in (3 * H, H, I, H))
s_qkv, s_fc1 = w_qkv.float().sum(1).contiguous(), w_fc1.float().sum(1).contiguous()
keep = torch.rand((R, T - 1), device=dev, generator=g) < 0.5
bits_all = ops.pack_mask(keep.to(torch.int64))


class Half:
    def __init__(self, rows, bits):
        self.rows, self.m, self.bits = rows, rows * T, bits.contiguous()
        m = self.m
        self.h = rnd(m, H); self.qkv = torch.empty((m, 3 * H), dtype=torch.bfloat16, device=dev)
        self.ctx = torch.empty((m, H), dtype=torch.bfloat16, device=dev); self.hx = torch.empty_like(self.ctx)
        self.inter = torch.empty((m, I), dtype=torch.bfloat16, device=dev); self.out = torch.empty_like(self.ctx)
        self.st1, self.st2 = ops.row_stats(self.h), ops.new_row_stats(m, H, dev)

    def qkv_(self):
        ops.gemm(self.h, w_qkv, b_qkv, L.AG_EPI_BIAS, BF, out=self.qkv, ln_stats=self.st1, ln_colsum=s_qkv, ln_eps=1e-12)

    def attn_(self):
        with L.on(dev):
            L.check(L.lib().ag_masked_attention(L.ptr(self.qkv), L.ptr(self.bits), L.ptr(self.ctx), self.rows, T, H, heads, 1, L.AG_MASK_VIT_MUL, 0, BF, L.stream()))

    def mlp_(self):
        ops.gemm(self.ctx, w_o, b_o, L.AG_EPI_BIAS_RESID, BF, resid=self.h, out=self.hx, stats_out=self.st2)
        ops.gemm(self.hx, w_fc1, b_fc1, L.AG_EPI_BIAS_GELU, BF, out=self.inter, ln_stats=self.st2, ln_colsum=s_fc1, ln_eps=1e-12)
        ops.gemm(self.inter, w_fc2, b_fc2, L.AG_EPI_BIAS_RESID, BF, resid=self.hx, out=self.out, stats_out=self.st1)


def wall(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n / LAYERS * 1e3


full = Half(R, bits_all)
def one_stream():
    for _ in range(LAYERS):
        full.qkv_(); full.attn_(); full.mlp_()
base = wall(one_stream)
print(json.dumps({"one_stream_ms_per_layer": round(base, 4), "rows": R}), flush=True)
a, b = Half(R // 2, bits_all[: R // 2]), Half(R - R // 2, bits_all[R // 2:])
def two_halves_one_stream():
    for _ in range(LAYERS):
        for x in (a, b): x.qkv_(); x.attn_(); x.mlp_()
print(json.dumps({"two_halves_one_stream_ms_per_layer": round(wall(two_halves_one_stream), 4)}), flush=True)
for c in [int(x) for x in os.environ.get("PP_SPLITS", "28,24").split(",")]:
    sg, st, ng, nt = ops.cu_partition_streams(dev, c)
    def piped():
        ev_q = {id(x): torch.cuda.Event() for x in (a, b)}; ev_a = {id(x): torch.cuda.Event() for x in (a, b)}
        cur = torch.cuda.current_stream()
        sg.wait_stream(cur); st.wait_stream(cur)
        with torch.cuda.stream(sg):
            for x in (a, b):
                x.qkv_(); ev_q[id(x)].record(sg)
        for _ in range(LAYERS):
            for x in (a, b):
                with torch.cuda.stream(st):
                    st.wait_event(ev_q[id(x)]); x.attn_(); ev_a[id(x)].record(st)
                with torch.cuda.stream(sg):
                    sg.wait_event(ev_a[id(x)]); x.mlp_(); x.qkv_(); ev_q[id(x)].record(sg)   # (the next layer's QKV of this half)
        cur.wait_stream(sg); cur.wait_stream(st)
    t_ = wall(piped)
    print(json.dumps({"cus_gemm": ng, "cus_attention": nt, "pipelined_ms_per_layer": round(t_, 4), "vs_one_stream": round(base / t_, 4)}), flush=True)
