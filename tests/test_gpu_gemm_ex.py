"""GPU parity of ag_gemm_ex (csrc/gemm_tn.hip: the GEMM of the under-filled launches) against float64 numpy on the same
bf16-rounded operands: the three operand orders of an nn.Linear's forward / dX / dW (what torch.autograd runs for
reference scripts/train_explainer.py:183-198, scripts/train_duo_explainer.py:180-198), every epilogue, split counts 1..8, at
M in {1 024, 1 576, 6 304} (BERT-base 8 x 128 tokens, ViT-base 8 x 197, one input x 32 masks x 197) with ragged M / N / Kc."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NT, NN, TN = (0, 0), (0, 1), (1, 1)
EX_STORE, EX_GELU_DUAL, EX_GELU_BWD, EX_SLABS = 0, 1, 2, 3


def _r(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.bfloat16)


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _gelu_grad(x):
    from scipy.special import erf
    return 0.5 * (1.0 + erf(x / np.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi)


def _operands(order, m, n, kc, seed, dev):
    """-> (device A, device B, float64 A [M,Kc], float64 B [N,Kc]) with A / B stored in the order's layout."""
    g = np.random.default_rng(seed)
    a = _r((g.standard_normal((m, kc)) * 1.1 + 0.1).astype(np.float32))
    b = _r((g.standard_normal((n, kc)) / np.sqrt(kc)).astype(np.float32))
    da = _dev(a.T if order[0] else a, dev)
    db = _dev(b.T if order[1] else b, dev)
    return g, da, db, a.astype(np.float64), b.astype(np.float64)


# (M, N, Kc): forward / dX shapes have M = rows; dW shapes have Kc = rows
SHAPES = {
    NT: [(1024, 768, 768), (1576, 2304, 768), (1576, 768, 3072), (6304, 768, 768), (1061, 200, 456)],
    NN: [(1024, 768, 3072), (1576, 3072, 768), (1576, 768, 2304), (6304, 768, 3072), (1061, 264, 200)],
    TN: [(768, 768, 1024), (2304, 768, 1576), (768, 3072, 1576), (3072, 768, 6304), (200, 264, 1061 - 5), (16, 192, 394)],
}


@pytest.mark.parametrize("order", [NT, NN, TN], ids=["NT", "NN", "TN"])
def test_gemm_ex_store_and_slabs(cuda_device, order):
    from autognothi_amd import ops
    for i, (m, n, kc) in enumerate(SHAPES[order]):
        g, da, db, a, b = _operands(order, m, n, kc, 10 + i, cuda_device)
        bias = g.standard_normal(n).astype(np.float32)
        ref = a @ b.T
        dbias = torch.from_numpy(bias).to(cuda_device)
        got32 = ops.gemm_ex(da, db, order, EX_STORE, bias=dbias, out_dtype=ops.F32).cpu().numpy()
        np.testing.assert_allclose(got32, ref + bias, rtol=2e-4, atol=2e-4, err_msg=f"{order} {m}x{n}x{kc} fp32 store")
        got16 = ops.gemm_ex(da, db, order, EX_STORE, bias=dbias, out_dtype=ops.BF16).float().cpu().numpy()
        np.testing.assert_allclose(got16, ref + bias, rtol=1e-2, atol=2e-2, err_msg=f"{order} {m}x{n}x{kc} bf16 store")
        nobias = ops.gemm_ex(da, db, order, EX_STORE, out_dtype=ops.F32).cpu().numpy()
        np.testing.assert_allclose(nobias, ref, rtol=2e-4, atol=2e-4)
        rec = ops.gemm_ex_splits(m, n, kc)
        assert 1 <= rec <= 8
        for s in sorted({1, 2, 3, rec, 8}):
            if s > max(1, (kc + 63) // 64):
                continue
            slabs = ops.gemm_ex(da, db, order, EX_SLABS, splits=s)
            assert slabs.shape == (s, m, n)
            tot = slabs.double().sum(0).cpu().numpy()
            np.testing.assert_allclose(tot, ref, rtol=2e-4, atol=2e-4, err_msg=f"{order} {m}x{n}x{kc} slabs s={s}")
            again = ops.gemm_ex(da, db, order, EX_SLABS, splits=s)
            assert torch.equal(slabs, again), "partial sums are not bit-reproducible"
        if i == 0:   # slab s holds exactly the contraction range [nk s / S, nk (s+1) / S) x 64
            s = 3
            slabs = ops.gemm_ex(da, db, order, EX_SLABS, splits=s).cpu().numpy()
            nk = (kc + 63) // 64
            for j in range(s):
                lo, hi = 64 * (nk * j // s), min(kc, 64 * (nk * (j + 1) // s))
                np.testing.assert_allclose(slabs[j], a[:, lo:hi] @ b[:, lo:hi].T, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("order", [NT, NN], ids=["NT", "NN"])
def test_gemm_ex_gelu_epilogues(cuda_device, order):
    from autognothi_amd import ops
    for i, (m, n, kc) in enumerate(SHAPES[order][:3] + SHAPES[order][4:]):
        g, da, db, a, b = _operands(order, m, n, kc, 40 + i, cuda_device)
        bias = g.standard_normal(n).astype(np.float32)
        pre = a @ b.T + bias
        dbias = torch.from_numpy(bias).to(cuda_device)
        got_pre, got_act = ops.gemm_ex(da, db, order, EX_GELU_DUAL, bias=dbias)
        gp = got_pre.float().cpu().numpy()
        np.testing.assert_allclose(gp, pre, rtol=1e-2, atol=2e-2)
        # the activation is gelu of the STORED (rounded) pre-activation, rounded to bf16 once
        np.testing.assert_allclose(got_act.float().cpu().numpy(), _gelu(gp.astype(np.float64)), rtol=8e-3, atol=2e-3)
        u = _r((g.standard_normal((m, n)) * 1.5).astype(np.float32))
        du = ops.gemm_ex(da, db, order, EX_GELU_BWD, aux=_dev(u, cuda_device)).float().cpu().numpy()
        want = (a @ b.T) * _gelu_grad(u.astype(np.float64))
        np.testing.assert_allclose(du, want, rtol=1e-2, atol=2e-2)


def test_gemm_ex_gelu_grad_accuracy(cuda_device):
    """the one-exponential gelu' against the erf form on a dense grid: identity product (A = I) isolates the epilogue."""
    from autognothi_amd import ops
    n = 1024
    eye = torch.eye(n, dtype=torch.bfloat16, device=cuda_device)
    xs = np.linspace(-10.0, 10.0, n * n, dtype=np.float32).reshape(n, n)
    u = _r(xs)
    got = ops.gemm_ex(eye, eye, NT, EX_GELU_BWD, aux=_dev(u, cuda_device)).float().cpu().numpy()
    want = np.eye(n) * _gelu_grad(u.astype(np.float64))
    d = np.abs(np.diag(got) - np.diag(want))
    assert d.max() <= 6e-3, d.max()      # bf16 output rounding (|gelu'| <= 1.13: half an ulp = 3.9e-3) + the tail fit


def test_gemm_ex_rejects_bad_arguments(cuda_device):
    from autognothi_amd import ops
    a = torch.zeros((64, 64), dtype=torch.bfloat16, device=cuda_device)
    with pytest.raises(RuntimeError):
        ops.gemm_ex(a, torch.zeros((12, 64), dtype=torch.bfloat16, device=cuda_device), NT, EX_STORE)    # N % 8
    with pytest.raises(TypeError):
        ops.gemm_ex(a.float(), a, NT, EX_STORE)
    with pytest.raises(RuntimeError):
        ops.gemm_ex(a.cpu(), a.cpu(), NT, EX_STORE)                                                       # no CPU fallback


@pytest.mark.parametrize("order", [NT, NN, TN], ids=["NT", "NN", "TN"])
def test_gemm_ex_staged_epilogue_equals_direct_stores(cuda_device, ag_knobs, order):
    """the shipped epilogue sends the tile through LDS and stores whole row segments; AG_GEMM_EX_EPI=0 stores straight from the
    accumulator layout: the same bits, every epilogue, ragged M / N edges included."""
    from autognothi_amd import ops
    shapes = {NT: [(1061, 200, 456), (1576, 768, 768)], NN: [(1061, 264, 200), (1576, 768, 768)], TN: [(200, 264, 1056), (768, 768, 1576)]}[order]

    def run_all(da, db, dbias, du, m, n, kc):
        outs = [ops.gemm_ex(da, db, order, EX_STORE, bias=dbias, out_dtype=ops.F32), ops.gemm_ex(da, db, order, EX_STORE, bias=dbias, out_dtype=ops.BF16),
                ops.gemm_ex(da, db, order, EX_SLABS, splits=3)]
        if order != TN:
            outs += list(ops.gemm_ex(da, db, order, EX_GELU_DUAL, bias=dbias)) + [ops.gemm_ex(da, db, order, EX_GELU_BWD, aux=du)]
        return outs

    for i, (m, n, kc) in enumerate(shapes):
        g, da, db, a, b = _operands(order, m, n, kc, 70 + i, cuda_device)
        dbias = torch.from_numpy(g.standard_normal(n).astype(np.float32)).to(cuda_device)
        du = _dev((g.standard_normal((m, n)) * 1.5).astype(np.float32), cuda_device)
        ag_knobs(AG_GEMM_EX_EPI=1)
        staged = run_all(da, db, dbias, du, m, n, kc)
        ag_knobs(AG_GEMM_EX_EPI=0)
        direct = run_all(da, db, dbias, du, m, n, kc)
        for x, y in zip(staged, direct):
            assert torch.equal(x, y)


@pytest.mark.parametrize("order", [NT, NN, TN], ids=["NT", "NN", "TN"])
def test_gemm_ex_group_equals_the_single_launches(cuda_device, order):
    """ag_gemm_ex_group (round 6: the dW products of two layers in one launch): every product of a group of 11 (two launches: 8 + 3;
    ragged shapes, a one-tile product, products whose first block is not a multiple of 8) against float64 and BIT FOR BIT against the
    same product as its own ag_gemm_ex launch with one contraction range."""
    from autognothi_amd import ops
    shapes = SHAPES[order] + [(768, 768, 1024), (136, 264, 512), (2304, 768, 1024), (768, 3072, 1024), (128, 128, 64), (392, 1000, 200)]
    if order == TN:
        shapes = [(m - m % 8, n, kc) for m, n, kc in shapes]
    else:
        shapes = [(m, n, kc - kc % 8) for m, n, kc in shapes]
    ops_ = [_operands(order, m, n, kc, 300 + i, cuda_device) for i, (m, n, kc) in enumerate(shapes)]
    for out_dtype in (ops.F32, ops.BF16):
        outs = ops.gemm_ex_group([(o[1], o[2]) for o in ops_], order, out_dtype)
        assert len(outs) == len(shapes)
        for (m, n, kc), o, got in zip(shapes, ops_, outs):
            single = ops.gemm_ex(o[1], o[2], order, EX_STORE, out_dtype=out_dtype)
            assert got.shape == (m, n) and torch.equal(got, single), (order, m, n, kc)
            ref = o[3] @ o[4].T
            tol = dict(rtol=2e-4, atol=2e-4) if out_dtype == ops.F32 else dict(rtol=1e-2, atol=2e-2)
            np.testing.assert_allclose(got.float().cpu().numpy(), ref, **tol, err_msg=f"{order} {m}x{n}x{kc}")


def test_colsum_bf16_group_equals_colsum_bf16(cuda_device):
    from autognothi_amd import ops
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn((m, n), generator=g).to(torch.bfloat16).to(cuda_device) for m, n in
          [(1024, 768), (1576, 3072), (1576, 2304), (7, 8), (1024, 3072), (300, 40), (64, 768)] * 3]
    outs = ops.colsum_bf16_group(xs)
    for x, o in zip(xs, outs):
        assert torch.equal(o, ops.colsum_bf16(x))
        np.testing.assert_allclose(o.cpu().numpy(), x.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
