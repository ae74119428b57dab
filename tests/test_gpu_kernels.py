"""GPU parity of the individual kernels against the numpy oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import shapley as osh
from oracle import transformer as otr
from util import golden, unpack

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


def _to_store(a, dtype, dev):
    t = torch.from_numpy(a).to(dev)
    return t.to(torch.bfloat16) if dtype == BF16 else t


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("m,n,k", [(1, 10, 768), (197, 576, 192), (130, 2304, 768), (394, 768, 3072), (8, 2, 768), (300, 132, 64)])
def test_gemm_epilogues(cuda_device, dtype, m, n, k):
    from autognothi_amd import _lib as L, ops
    g = np.random.default_rng(m * 131 + n)
    a = g.standard_normal((m, k)).astype(np.float32)
    w = (g.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = g.standard_normal(n).astype(np.float32)
    r = g.standard_normal((m, n)).astype(np.float32)
    if dtype == BF16:
        a, w = _bf16_round(a), _bf16_round(w)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    A, W = _to_store(a, dtype, cuda_device), _to_store(w, dtype, cuda_device)
    if dtype == BF16:
        r = _bf16_round(r)
    B, R = torch.from_numpy(b).to(cuda_device), _to_store(r, dtype, cuda_device)
    tol = dict(rtol=1e-5, atol=2e-5) if dtype == F32 else dict(rtol=1e-2, atol=2e-2)
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS_F32, dtype).cpu().numpy()
    np.testing.assert_allclose(out, ref, **(dict(rtol=1e-5, atol=2e-5)))  # fp32 accumulate of exact inputs in both modes
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS, dtype).float().cpu().numpy()
    np.testing.assert_allclose(out, ref, **tol)
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS_GELU, dtype).float().cpu().numpy()
    np.testing.assert_allclose(out, otr.gelu(ref.astype(np.float32)), **tol)
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS_TANH, dtype).float().cpu().numpy()
    np.testing.assert_allclose(out, np.tanh(ref), **tol)
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS_RESID, dtype, resid=R).float().cpu().numpy()
    np.testing.assert_allclose(out, ref + r, **(dict(rtol=1e-5, atol=3e-5) if dtype == F32 else tol))


def test_gemm_strided_rows_and_shared_residual(cuda_device):
    """lda/ldc/ldr strides (CLS-only last layer) and the layer-0 residual shared by K masked rows."""
    from autognothi_amd import _lib as L, ops
    g = np.random.default_rng(9)
    rows, t, h, share = 12, 5, 128, 4
    a = g.standard_normal((rows, t, h)).astype(np.float32)
    w = (g.standard_normal((h, h)) / np.sqrt(h)).astype(np.float32)
    b = g.standard_normal(h).astype(np.float32)
    res = g.standard_normal((rows // share, t, h)).astype(np.float32)
    A, W, B, R = [torch.from_numpy(x).to(cuda_device) for x in (a, w, b, res)]
    # all tokens, residual row = ((m/T)/share)*T + m%T
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS_RESID, F32, m=rows * t, resid=R, rows_per_seq=t, resid_share=share).cpu().numpy()
    ref = a.reshape(-1, h) @ w.T + b + np.repeat(res, share, axis=0).reshape(-1, h)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=3e-5)
    # token 0 only: lda = ldr = T*H, output written in place into [rows,T,H] with ldc = T*H
    dst = torch.zeros((rows, t, h), device=cuda_device)
    ops.gemm(A, W, B, L.AG_EPI_BIAS_RESID, F32, m=rows, lda=t * h, resid=R, ldr=t * h, rows_per_seq=1, resid_share=share,
             out=dst, ldc=t * h)
    ref0 = a[:, 0] @ w.T + b + np.repeat(res[:, 0], share, axis=0)
    np.testing.assert_allclose(dst[:, 0].cpu().numpy(), ref0, rtol=1e-5, atol=3e-5)
    assert float(dst[:, 1:].abs().max()) == 0.0


def test_gemm_rejects_bad_k(cuda_device):
    from autognothi_amd import _lib as L, ops
    a = torch.zeros((4, 102), device=cuda_device)
    w = torch.zeros((8, 102), device=cuda_device)
    with pytest.raises(RuntimeError, match="multiple"):
        ops.gemm(a, w, None, L.AG_EPI_BIAS_F32, F32)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("h", [96, 128, 20, 192, 768, 1024])   # <= 128: the 8-lanes-per-row kernel (96 = the LTT side width)
def test_layernorm(cuda_device, dtype, h):
    from autognothi_amd import ops
    g = np.random.default_rng(h)
    x = (g.standard_normal((37, h)) * 3 + 1).astype(np.float32)
    sd = {"ln.weight": g.standard_normal(h).astype(np.float32), "ln.bias": g.standard_normal(h).astype(np.float32)}
    for eps in (1e-12, 1e-5):
        ref = otr.layer_norm(x, sd, "ln", eps)
        ys, yf = ops.layernorm(torch.from_numpy(x).to(cuda_device), torch.from_numpy(sd["ln.weight"]).to(cuda_device),
                               torch.from_numpy(sd["ln.bias"]).to(cuda_device), eps, dtype, want_f32=True)
        np.testing.assert_allclose(yf.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        xb = _bf16_round(x)  # bf16 residual stream as LayerNorm input
        _, yb = ops.layernorm(torch.from_numpy(xb).to(cuda_device).to(torch.bfloat16), torch.from_numpy(sd["ln.weight"]).to(cuda_device),
                              torch.from_numpy(sd["ln.bias"]).to(cuda_device), eps, dtype, want_f32=True)
        np.testing.assert_allclose(yb.cpu().numpy(), otr.layer_norm(xb, sd, "ln", eps), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ys.float().cpu().numpy(), ref, rtol=1e-2 if dtype == BF16 else 1e-5, atol=2e-2 if dtype == BF16 else 1e-5)


def _attention_case(cuda_device, dtype, mode, t, heads, rows, share, n_query, seed, d=64):
    from autognothi_amd import ops
    h = heads * d
    g = np.random.default_rng(seed)
    src = rows // share
    qkv = g.standard_normal((src, t, 3 * h)).astype(np.float32)
    if dtype == BF16:
        qkv = _bf16_round(qkv)
    mask = g.integers(0, 2, size=(rows, t - 1), dtype=np.int64)
    mask[0] = 0          # all players off (only CLS on)
    mask[-1] = 1         # all on
    sd = {}
    eye = np.eye(h, dtype=np.float32)
    for i, nm in enumerate(("query", "key", "value")):
        sd[f"a.{nm}.weight"], sd[f"a.{nm}.bias"] = eye, np.zeros(h, dtype=np.float32)
    q, k, v = qkv[..., :h], qkv[..., h:2 * h], qkv[..., 2 * h:]
    # oracle attention on (q,k,v) directly: feed u through identity projections per stream
    def heads_(x):
        return x.reshape(x.shape[0], t, heads, d).transpose(0, 2, 1, 3)
    qh, kh, vh = [heads_(np.repeat(x, share, axis=0)) for x in (q, k, v)]
    s = (qh @ kh.transpose(0, 1, 3, 2)) / np.float32(np.sqrt(d))
    m = otr.prepend_cls(mask).astype(np.float32)[:, None, None, :]
    s = s * m if mode == 0 else s + (1 - m) * otr.F32_MIN
    ref = (otr.softmax(s) @ vh).transpose(0, 2, 1, 3).reshape(rows, t, h)
    bits = ops.pack_mask(torch.from_numpy(mask).to(cuda_device))
    QKV = torch.from_numpy(qkv).to(cuda_device)
    QKV = QKV.to(torch.bfloat16) if dtype == BF16 else QKV
    out = ops.masked_attention(QKV, bits, rows, t, h, heads, share, mode, dtype, n_query=n_query).float().cpu().numpy()
    nq = n_query or t
    tol = dict(rtol=1e-4, atol=2e-5) if dtype == F32 else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(out[:, :nq], ref[:, :nq], **tol)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("t,heads,rows,share,n_query", [(197, 3, 4, 1, 0), (197, 2, 6, 3, 0), (128, 2, 4, 2, 0),
                                                        (197, 2, 4, 1, 1), (33, 1, 2, 1, 0), (512, 1, 2, 1, 0), (64, 1, 2, 2, 1)])
def test_masked_attention(cuda_device, dtype, mode, t, heads, rows, share, n_query):
    _attention_case(cuda_device, dtype, mode, t, heads, rows, share, n_query, seed=t * 7 + heads)


@pytest.mark.parametrize("t,heads,rows,share,n_query", [(197, 12, 56, 1, 0), (197, 12, 64, 8, 0), (100, 6, 96, 2, 0), (230, 4, 130, 1, 1)])
def test_masked_attention_many_items(cuda_device, t, heads, rows, share, n_query):
    """many (row, head) workgroups with every mask density: T = 197 runs the 7-block unrolled ViT kernel (masked keys =
    zeroed K rows), other T the runtime loop; shared layer-0 rows; CLS-only queries."""
    _attention_case(cuda_device, BF16, 0, t, heads, rows, share, n_query, seed=t + rows)


@pytest.mark.parametrize("t,heads,rows,share,n_query", [(197, 12, 24, 1, 0), (197, 12, 56, 1, 0), (197, 12, 64, 8, 0), (193, 12, 43, 1, 0), (200, 16, 50, 2, 0),
                                                        (197, 3, 260, 1, 0), (197, 12, 56, 1, 1), (197, 12, 43, 1, 40)])
def test_masked_attention_stream3(cuda_device, ag_knobs, t, heads, rows, share, n_query):
    """attn_stream3_kernel (one K/V stream per CU, two wave teams, three LDS images; ViT rows of 193-200 tokens): against the oracle
    and BIT FOR BIT against the workgroup-per-item kernel (same block bodies in the same order), with one to four items per
    workgroup (odd and even counts, workgroups with a different number of items), shared layer-0 rows, T at both ends of the range, the first
    n_query tokens as queries (the CLS-only last layer: query waves without a query only stage)."""
    from autognothi_amd import ops
    ag_knobs(AG_ATTN_STREAM3=1, AG_ATTN_STREAM3_MIN=1)
    _attention_case(cuda_device, BF16, 0, t, heads, rows, share, n_query, seed=t + rows)
    nq = n_query or t
    h = heads * 64
    g = torch.Generator(device=cuda_device); g.manual_seed(rows)
    qkv = torch.randn((rows // share, t, 3 * h), device=cuda_device, generator=g).to(torch.bfloat16)
    keep = (torch.rand((rows, t - 1), device=cuda_device, generator=g) < 0.5).to(torch.int64)
    bits = ops.pack_mask(keep)
    new = ops.masked_attention(qkv, bits, rows, t, h, heads, share, 0, BF16, n_query=n_query)[:, :nq]
    again = ops.masked_attention(qkv, bits, rows, t, h, heads, share, 0, BF16, n_query=n_query)[:, :nq]
    ag_knobs(AG_ATTN_STREAM3=0)
    old = ops.masked_attention(qkv, bits, rows, t, h, heads, share, 0, BF16, n_query=n_query)[:, :nq]
    assert torch.equal(new.contiguous().view(torch.int16), old.contiguous().view(torch.int16))
    assert torch.equal(new.contiguous().view(torch.int16), again.contiguous().view(torch.int16))


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("d,t,heads,rows,share", [(8, 197, 12, 3, 1), (8, 128, 3, 4, 2), (16, 65, 2, 2, 1), (32, 197, 1, 2, 1)])
def test_masked_attention_narrow_heads(cuda_device, dtype, mode, d, t, heads, rows, share):
    """head dims other than 64 (the 8-wide heads of the LTT side network, reference models/ltt_vit.py:383-394)."""
    _attention_case(cuda_device, dtype, mode, t, heads, rows, share, 0, seed=d * 13 + t, d=d)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("m,n,k", [(394, 96, 768), (197, 24, 192), (260, 288, 96), (130, 96, 384), (64, 40, 24),
                                   (2100, 96, 768), (70001, 96, 768), (3000, 128, 256), (2500, 64, 1024), (2049, 32, 128)])
def test_gemm_ladder_shapes(cuda_device, dtype, m, n, k):
    """LTT ladder GEMMs: K not a multiple of the 128-byte LDS slice (zero-sourced K tail) and the
    side = side + gelu(Linear(hidden)) epilogue (reference models/ltt_vit.py:431)."""
    from autognothi_amd import _lib as L, ops
    g = np.random.default_rng(m + 7 * n + k)
    a = g.standard_normal((m, k)).astype(np.float32)
    w = (g.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = g.standard_normal(n).astype(np.float32)
    r = g.standard_normal((m, n)).astype(np.float32)
    if dtype == BF16:
        a, w, r = _bf16_round(a), _bf16_round(w), _bf16_round(r)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    A, W, R = _to_store(a, dtype, cuda_device), _to_store(w, dtype, cuda_device), _to_store(r, dtype, cuda_device)
    B = torch.from_numpy(b).to(cuda_device)
    tol = dict(rtol=1e-5, atol=3e-5) if dtype == F32 else dict(rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(ops.gemm(A, W, B, L.AG_EPI_BIAS_F32, dtype).cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS_GELU_ADD, dtype, resid=R).float().cpu().numpy()
    np.testing.assert_allclose(out, otr.gelu(ref.astype(np.float32)) + r, **tol)
    # (bf16, M >= 2048, N in {32, 64, 96, 128}: the LDS-resident map kernel of csrc/side_mlp.hip, also for the first ladder layer's
    # plain GELU epilogue)
    out = ops.gemm(A, W, B, L.AG_EPI_BIAS_GELU, dtype).float().cpu().numpy()
    np.testing.assert_allclose(out, otr.gelu(ref.astype(np.float32)), **tol)
    # NaN/Inf in the bytes after a row must not leak into the K tail: the tail chunks come from a zero buffer
    pad = torch.full((m, k + 40), float("nan"), device=cuda_device, dtype=A.dtype)
    pad[:, :k] = A
    out = ops.gemm(pad, W, B, L.AG_EPI_BIAS_F32, dtype, m=m, lda=k + 40).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=2e-5)


def test_shapley_reductions(cuda_device):
    from autognothi_amd import ops
    g = golden("shapley_fns.npz")
    dev = cuda_device
    for tag in ("vit", "bert"):
        b, k, p, c = [int(x) for x in g[f"{tag}_dims"]]
        pred, grand, null = [torch.from_numpy(g[f"{tag}_norm_{n}"]).to(dev) for n in ("pred", "grand", "null")]
        phi = ops.shapley_normalize(pred, grand, null).cpu().numpy()
        np.testing.assert_allclose(phi, g[f"{tag}_norm_out"][:, 1:, :].transpose(0, 2, 1), rtol=0, atol=1e-5)
        raw = ops.shapley_normalize(pred, None, None, normalize=False).cpu().numpy()
        np.testing.assert_array_equal(raw, g[f"{tag}_norm_pred"][:, 1:, :].transpose(0, 2, 1))
        # backward of normalise: compare with finite structure from the oracle formula
        dphi = np.random.default_rng(1).standard_normal(phi.shape).astype(np.float32)
        dpred = ops.shapley_normalize_bwd(torch.from_numpy(dphi).to(dev), p + 1).cpu().numpy()
        want = np.zeros((b, p + 1, c), dtype=np.float64)
        want[:, 1:, :] = dphi.transpose(0, 2, 1)
        want -= dphi.sum(axis=2)[:, None, :] / (p + 1)
        np.testing.assert_allclose(dpred, want, rtol=1e-5, atol=1e-6)

        mask = unpack(g[f"{tag}_loss_mask"], p).reshape(b * k, p)
        bits = ops.pack_mask(torch.from_numpy(mask).to(dev))
        loss, dphi = ops.shapley_loss(bits, torch.from_numpy(g[f"{tag}_loss_v0"]).to(dev), torch.from_numpy(g[f"{tag}_loss_vs"]).to(dev),
                                      torch.from_numpy(g[f"{tag}_loss_phi"]).to(dev), b, k)
        np.testing.assert_allclose(loss.cpu().numpy()[0], g[f"{tag}_loss_out"][0], rtol=1e-5)
        np.testing.assert_allclose(dphi.cpu().numpy(), g[f"{tag}_loss_dphi"], rtol=1e-4, atol=3e-6)  # fp32 sums with cancellation

        kl, dcur = ops.kl_loss(torch.from_numpy(g[f"{tag}_kl_ref"]).to(dev), torch.from_numpy(g[f"{tag}_kl_cur"]).to(dev))
        np.testing.assert_allclose(kl.cpu().numpy()[0], g[f"{tag}_kl_out"][0], rtol=1e-5, atol=1e-7)
        cur = torch.from_numpy(g[f"{tag}_kl_cur"]).double().requires_grad_(True)
        ref = torch.from_numpy(g[f"{tag}_kl_ref"]).double()
        l = torch.nn.functional.kl_div(torch.log_softmax(ref, -1), torch.softmax(cur, -1), reduction="batchmean")
        l.backward()
        np.testing.assert_allclose(dcur.cpu().numpy(), cur.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_cpu_tensors_rejected():
    from autognothi_amd import _lib as L, ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 64), torch.zeros(4, 64), None, L.AG_EPI_BIAS_F32, F32)


def test_gemm_layernorm_fold(cuda_device, ag_knobs):
    """Linear(LayerNorm(x)) through the folded epilogue (row statistics + gamma-scaled weights) vs the oracle, and
    the producer side (statistics accumulated by the epilogue that writes the rows)."""
    from autognothi_amd import _lib as L, ops
    ag_knobs(AG_GEMM_BIG_MIN_TILES=1)     # the 2048 x 768 producer is 24 tiles: pin the ring kernel
    g = np.random.default_rng(21)
    m, h, n = 2048, 768, 2304
    x = _bf16_round((g.standard_normal((m, h)) * 1.5 + 0.3).astype(np.float32))
    w = (g.standard_normal((n, h)) / np.sqrt(h)).astype(np.float32)
    b = g.standard_normal(n).astype(np.float32)
    gamma = (1 + 0.1 * g.standard_normal(h)).astype(np.float32)
    beta = (0.1 * g.standard_normal(h)).astype(np.float32)
    eps = 1e-12
    ref = otr.layer_norm(x, {"ln.weight": gamma, "ln.bias": beta}, "ln", eps).astype(np.float64) @ w.astype(np.float64).T + b
    dev = cuda_device
    X = torch.from_numpy(x).to(dev).to(torch.bfloat16)
    wf = torch.from_numpy(w * gamma[None, :]).to(dev).to(torch.bfloat16)
    bias_f = torch.from_numpy((b + w @ beta).astype(np.float32)).to(dev)
    colsum = wf.float().sum(dim=1).contiguous()
    stats = ops.row_stats(X)
    np.testing.assert_allclose(ops.reduce_row_stats(stats, m, h).cpu().numpy()[:, 0], x.sum(1), rtol=1e-5, atol=1e-3)
    out = ops.gemm(X, wf, bias_f, L.AG_EPI_BIAS, BF16, ln_stats=stats, ln_colsum=colsum, ln_eps=eps).float().cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=2e-2, atol=3e-2)
    # producer: a residual GEMM that writes rows and accumulates their (sum, sumsq)
    a = _bf16_round(g.standard_normal((m, h)).astype(np.float32))
    w2 = _bf16_round((g.standard_normal((h, h)) / np.sqrt(h)).astype(np.float32))
    st_out = ops.new_row_stats(m, h, dev)
    hx = ops.gemm(torch.from_numpy(a).to(dev).to(torch.bfloat16), torch.from_numpy(w2).to(dev).to(torch.bfloat16), None,
                  L.AG_EPI_BIAS_RESID, BF16, resid=X, stats_out=st_out)
    hxf = hx.float().cpu().numpy()
    tot = ops.reduce_row_stats(st_out, m, h).cpu().numpy()
    np.testing.assert_allclose(tot[:, 0], hxf.sum(1), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(tot[:, 1], (hxf.astype(np.float64) ** 2).sum(1), rtol=1e-4)
    # the folded path is refused where it is not implemented (small / fp32 GEMMs) instead of silently ignored
    with pytest.raises(RuntimeError, match="folding"):
        ops.gemm(X[:64], wf, bias_f, L.AG_EPI_BIAS, BF16, ln_stats=ops.row_stats(X[:64]), ln_colsum=colsum)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("t,heads,rows,cls_only", [(128, 2, 6, 0), (128, 12, 5, 1), (65, 1, 4, 0)])
def test_token_pruning_blocks(cuda_device, dtype, t, heads, rows, cls_only):
    """BERT token pruning: compaction plan (popcount scan + source table), row gather, mask-free varlen attention ==
    additive-mask attention of the visible tokens (reference models/vanilla_bert.py:523: masked keys get weight 0)."""
    import ctypes as C
    from autognothi_amd import _lib as L, ops
    dev = cuda_device
    h, d = heads * 64, 64
    g = np.random.default_rng(t + heads)
    mask = g.integers(0, 2, size=(rows, t - 1), dtype=np.int64)
    mask[0] = 0
    mask[-1] = 1
    bits = ops.pack_mask(torch.from_numpy(mask).to(dev))
    cu = torch.empty(rows + 1, dtype=torch.int32, device=dev)
    src = torch.empty(rows * t, dtype=torch.int32, device=dev)
    L.check(L.lib().ag_seq_compact_plan(L.ptr(bits), rows, t, L.ptr(cu), L.ptr(src), L.stream()))
    full = otr.prepend_cls(mask)
    counts = full.sum(1)
    want_cu = np.concatenate([[0], np.cumsum(counts)])
    np.testing.assert_array_equal(cu.cpu().numpy(), want_cu)
    n = int(want_cu[-1])
    want_src = np.concatenate([r * t + np.nonzero(full[r])[0] for r in range(rows)])
    np.testing.assert_array_equal(src.cpu().numpy()[:n], want_src)
    # gather the visible tokens of a [rows, t, 3h] qkv
    qkv = g.standard_normal((rows, t, 3 * h)).astype(np.float32)
    if dtype == BF16:
        qkv = _bf16_round(qkv)
    QKV = _to_store(qkv, dtype, dev)
    packed = torch.empty((n, 3 * h), dtype=QKV.dtype, device=dev)
    L.check(L.lib().ag_gather_rows(L.ptr(QKV), 3 * h, L.ptr(src), L.ptr(packed), 3 * h, n, 3 * h, dtype, None, L.stream()))
    np.testing.assert_array_equal(packed.float().cpu().numpy(), qkv.reshape(rows * t, 3 * h)[want_src])
    ctx = torch.zeros((n, h), dtype=QKV.dtype, device=dev)
    L.check(L.lib().ag_masked_attention_varlen(L.ptr(packed), L.ptr(cu), L.ptr(ctx), rows, t, h, heads, cls_only, dtype, L.stream()))
    got = ctx.float().cpu().numpy()
    # oracle: additive-mask attention on the unpacked rows, compared on the visible tokens
    q, k, v = qkv[..., :h], qkv[..., h:2 * h], qkv[..., 2 * h:]
    hd = lambda x: x.reshape(rows, t, heads, d).transpose(0, 2, 1, 3)  # noqa: E731
    s = (hd(q) @ hd(k).transpose(0, 1, 3, 2)) / np.float32(8.0)
    m = full.astype(np.float32)[:, None, None, :]
    ref = (otr.softmax(s + (1 - m) * otr.F32_MIN) @ hd(v)).transpose(0, 2, 1, 3).reshape(rows * t, h)[want_src]
    tol = dict(rtol=1e-4, atol=2e-5) if dtype == F32 else dict(rtol=2e-2, atol=2e-2)
    sel = want_cu[:-1] if cls_only else np.arange(n)
    np.testing.assert_allclose(got[sel], ref[sel], **tol)


def test_empty_inputs(cuda_device):
    """Zero-row inputs are legal everywhere the reference would produce empty tensors (e.g. mask_shapley_new(0, P) returns
    [0, P]); nothing may launch an empty grid."""
    from autognothi_amd import _lib as L, ops
    dev = cuda_device
    rng = ops.DeviceMT19937(dev, 1)
    before = rng.raw(4).cpu().numpy()
    rng.seed(1)
    mi, mb = ops.mask_shapley_new(rng, 0, 196)
    assert tuple(mi.shape) == (0, 196) and mb.shape[0] == 0
    np.testing.assert_array_equal(rng.raw(4).cpu().numpy(), before)            # no draws were consumed
    a = torch.zeros((0, 768), device=dev)
    w = torch.zeros((16, 768), device=dev)
    assert tuple(ops.gemm(a, w, None, L.AG_EPI_BIAS_F32, F32, m=0).shape) == (0, 16)
    bits = torch.zeros((0, 7), dtype=torch.int32, device=dev)
    qkv = torch.zeros((0, 197, 3 * 64), device=dev)
    assert ops.masked_attention(qkv, bits, 0, 197, 64, 1, 1, 0, F32).shape[0] == 0
    assert tuple(ops.pack_mask(torch.zeros((0, 196), dtype=torch.int64, device=dev)).shape) == (0, 7)
    x = torch.zeros((0, 192), device=dev)
    y, _ = ops.layernorm(x, torch.ones(192, device=dev), torch.zeros(192, device=dev), 1e-12, F32, rows=0, ldx=192)
    assert y.shape[0] == 0


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_dynamic_row_counts(cuda_device, dtype, ag_knobs):
    """d_rows (ops: rows_dev): launches sized for an upper bound, the actual row count read from device memory (the packed token
    count of a pruned BERT forward never visits the host): rows below the count are computed exactly as by an exact-size
    launch, rows at or above it are left untouched — for the 128-tile GEMM, the ring GEMM, LayerNorm and the row gather."""
    from autognothi_amd import _lib as L, ops
    ag_knobs(AG_GEMM_BIG_MIN_TILES=1)     # 9 x 4 tiles: pin the ring kernel for the large-M case
    dev = cuda_device
    g = np.random.default_rng(4)
    upper, actual, k, n = 2304, 1237, 768, 776
    a = g.standard_normal((upper, k)).astype(np.float32)
    w = (g.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = g.standard_normal(n).astype(np.float32)
    A, W, B = _to_store(a, dtype, dev), _to_store(w, dtype, dev), torch.from_numpy(b).to(dev)
    cnt = torch.tensor([actual], dtype=torch.int32, device=dev)
    sentinel = 768.0      # (exactly representable in bf16)

    for m_up in (upper, 600):         # ring kernel (bf16, M >= 1024) and the 128-tile kernel
        act = min(actual, m_up) if m_up == upper else 333
        cnt.fill_(act)
        want = ops.gemm(A[:act], W, B, L.AG_EPI_BIAS_GELU, dtype).float()
        out = torch.full((m_up, n), sentinel, dtype=want.dtype if dtype == F32 else torch.bfloat16, device=dev)
        ops.gemm(A[:m_up], W, B, L.AG_EPI_BIAS_GELU, dtype, out=out, rows_dev=cnt)
        assert torch.equal(out[:act].float(), want)
        assert bool((out[act:].float() == sentinel).all())
    cnt.fill_(actual)
    x = _to_store(a, dtype, dev)
    gam, bet = torch.from_numpy(g.standard_normal(k).astype(np.float32)).to(dev), torch.from_numpy(g.standard_normal(k).astype(np.float32)).to(dev)
    want, _ = ops.layernorm(x[:actual], gam, bet, 1e-12, dtype)
    got, _ = ops.layernorm(x, gam, bet, 1e-12, dtype, rows_dev=cnt)
    assert torch.equal(got[:actual], want)
    idx = torch.from_numpy(g.permutation(upper).astype(np.int32)).to(dev)
    want = ops.gather_rows(x, idx, actual, dtype)
    dst = ops.gather_rows(x, idx, upper, dtype, rows_dev=cnt)
    assert torch.equal(dst[:actual], want)
    # there is no mode to leak: a call without rows_dev sees its full row count
    full = ops.gemm(A, W, B, L.AG_EPI_BIAS, dtype)
    assert bool(torch.isfinite(full.float()).all()) and float(full[actual:].float().abs().max()) > 0


@pytest.mark.parametrize("post_ln", [False, True])
@pytest.mark.parametrize("m,h,i", [(1000, 96, 384), (257, 96, 384), (5000, 128, 256), (300, 64, 256), (33, 32, 128), (70000, 96, 384)])
def test_side_mlp_fused(cuda_device, post_ln, m, h, i):
    """ag_side_mlp (LN + fc1 + GELU + fc2 + residual of a narrow layer in one kernel, csrc/side_mlp.hip) against the numpy
    oracle's layer arithmetic: ViT block x + fc2(gelu(fc1(LN(x)))) (models/vanilla_vit.py:373-376) and BERT block
    LN(x + fc2(gelu(fc1(x)))) (models/vanilla_bert.py:576-577,:601-603); ragged row counts, every supported width."""
    from autognothi_amd import _lib as L, ops
    assert L.lib().ag_side_mlp_supported(h, i, BF16) == 1
    g = np.random.default_rng(m + h + i)
    x = _bf16_round((g.standard_normal((m, h)) * 1.3 + 0.2).astype(np.float32))
    w1 = _bf16_round((g.standard_normal((i, h)) / np.sqrt(h)).astype(np.float32))
    w2 = _bf16_round((g.standard_normal((h, i)) / np.sqrt(i)).astype(np.float32))
    b1, b2 = g.standard_normal(i).astype(np.float32) * 0.3, g.standard_normal(h).astype(np.float32) * 0.3
    gam, bet = (1 + 0.1 * g.standard_normal(h)).astype(np.float32), (0.1 * g.standard_normal(h)).astype(np.float32)
    sd = {"ln.weight": gam, "ln.bias": bet}
    eps = 1e-12
    if post_ln:
        f = _bf16_round(otr.gelu((x.astype(np.float64) @ w1.astype(np.float64).T + b1).astype(np.float32)))
        ref = otr.layer_norm((f.astype(np.float64) @ w2.astype(np.float64).T + b2 + x).astype(np.float32), sd, "ln", eps)
    else:
        u = _bf16_round(otr.layer_norm(x, sd, "ln", eps))
        f = _bf16_round(otr.gelu((u.astype(np.float64) @ w1.astype(np.float64).T + b1).astype(np.float32)))
        ref = f.astype(np.float64) @ w2.astype(np.float64).T + b2 + x
    dev = cuda_device
    t = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
    out = ops.side_mlp(t(x).to(torch.bfloat16), t(w1).to(torch.bfloat16), t(b1), t(w2).to(torch.bfloat16), t(b2), t(gam), t(bet), eps, post_ln)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1e-2, atol=3e-2)
    if not post_ln:   # Identity LayerNorm (explainer side layer 0: repl_norm_1 / no LN parameters)
        f = _bf16_round(otr.gelu((x.astype(np.float64) @ w1.astype(np.float64).T + b1).astype(np.float32)))
        ref = f.astype(np.float64) @ w2.astype(np.float64).T + b2 + x
        out = ops.side_mlp(t(x).to(torch.bfloat16), t(w1).to(torch.bfloat16), t(b1), t(w2).to(torch.bfloat16), t(b2), None, None, eps, False)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1e-2, atol=3e-2)
    assert L.lib().ag_side_mlp_supported(24, 96, BF16) == 0 and L.lib().ag_side_mlp_supported(96, 384, F32) == 0


@pytest.mark.parametrize("m,h,n,pre,res,post", [(1000, 96, 288, True, False, False), (257, 96, 288, False, False, False),
                                                (1000, 96, 96, False, True, False), (3001, 96, 96, False, True, True),
                                                (500, 128, 384, True, False, False), (300, 64, 64, False, True, True),
                                                (90, 32, 96, True, True, False), (70000, 96, 288, True, False, False)])
def test_side_linear_fused(cuda_device, m, h, n, pre, res, post):
    """ag_side_linear: LN_post(resid + W . LN_pre(x) + b) for the attention half of a narrow layer — LN1 + QKV (ViT
    models/vanilla_vit.py:369,:437-441), QKV (BERT), out-proj + residual (ViT :372,:477), out-proj + residual + LayerNorm
    (BERT models/vanilla_bert.py:557-559) — against the numpy oracle; the output feature permutation inside the kernel
    (16-byte stores) must be invisible."""
    from autognothi_amd import _lib as L, ops
    assert L.lib().ag_side_linear_supported(h, n, 1 if post else 0, BF16) == 1
    g = np.random.default_rng(m + h + n)
    x = _bf16_round((g.standard_normal((m, h)) * 1.3 + 0.2).astype(np.float32))
    w = _bf16_round((g.standard_normal((n, h)) / np.sqrt(h)).astype(np.float32))
    b = (g.standard_normal(n) * 0.3).astype(np.float32)
    r = _bf16_round(g.standard_normal((m, n)).astype(np.float32))
    g0, b0 = (1 + 0.1 * g.standard_normal(h)).astype(np.float32), (0.1 * g.standard_normal(h)).astype(np.float32)
    g1, b1 = (1 + 0.1 * g.standard_normal(n)).astype(np.float32), (0.1 * g.standard_normal(n)).astype(np.float32)
    eps = 1e-12
    u = _bf16_round(otr.layer_norm(x, {"ln.weight": g0, "ln.bias": b0}, "ln", eps)) if pre else x
    ref = u.astype(np.float64) @ w.astype(np.float64).T + b
    if res:
        ref = ref + r
    if post:
        ref = otr.layer_norm(ref.astype(np.float32), {"ln.weight": g1, "ln.bias": b1}, "ln", eps)
    t = lambda a: torch.from_numpy(a).to(cuda_device)   # noqa: E731
    out = ops.side_linear(t(x).to(torch.bfloat16), t(w).to(torch.bfloat16), t(b), (t(g0), t(b0)) if pre else None,
                          t(r).to(torch.bfloat16) if res else None, (t(g1), t(b1)) if post else None, eps)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1e-2, atol=3e-2)
    assert L.lib().ag_side_linear_supported(96, 288, 1, BF16) == 0      # a LayerNorm over 288 outputs is not built


@pytest.mark.parametrize("m,n", [(1576, 768), (1576, 3072), (12608, 2304), (37, 10), (8, 2), (5000, 70), (1, 768), (333, 4099)])
def test_colsum_bias_gradients(cuda_device, m, n):
    """ag_colsum_f32 (bias gradients of the training step): one launch, fixed summation tree — against float64 sums, with
    the accumulate form, unaligned widths (num_labels = 10 / 2), and bit-identical from launch to launch."""
    from autognothi_amd import ops
    g = torch.Generator().manual_seed(m * 31 + n)
    x = torch.randn((m, n), generator=g)
    xd = x.to(cuda_device)
    want = x.double().sum(0)
    got = ops.colsum(xd)
    tol = 2e-6 * float(x.abs().double().sum(0).max()) + 1e-6
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=0, atol=tol)
    for _ in range(3):
        assert torch.equal(ops.colsum(xd), got)
    acc = torch.full((n,), 2.0, device=cuda_device)
    ops.colsum(xd, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().double().numpy(), want.numpy() + 2.0, rtol=0, atol=tol)
