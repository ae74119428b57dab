"""ag_gemm_resid_split: the bias + residual GEMM with its under-filled tail round of 256 x 256 tiles computed as contraction ranges side
by side (fp32 partial tiles) and finished by a row kernel — against float64, against the unsplit launch, and its slab statistics
against sums over what it stored.  Shapes = fc2 of ViT-base / BERT-base at the reference's own batch sizes (1, 4, 5 inputs x K = 32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = 1
TOL = dict(rtol=1e-2, atol=2e-2)      # bf16 storage of the result


def _r(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


def _dev(a, dev):
    return torch.from_numpy(a).to(dev).to(torch.bfloat16)


# rows: 1 input x 32 masks x 197 tokens (75 tiles: all tail, 3 ranges) | 4 inputs (one round + 42 tiles, 6 ranges) | a tail of 87 tiles (2 ranges)
# | BERT-base 1 x 32 x 128 (48 tiles, 4 ranges) | a ragged last panel
@pytest.mark.parametrize("m,n,k", [(6304, 768, 3072), (25216, 768, 3072), (29184, 768, 3072), (4096, 768, 3072), (22000, 768, 3072)])
def test_resid_split_vs_float64_and_unsplit(cuda_device, m, n, k):
    from autognothi_amd import _lib as L, ops
    dev = cuda_device
    with torch.cuda.device(dev):
        need = ops.gemm_resid_split_scratch_bytes(m, n, k)
    if need == 0:
        pytest.skip("this device's CU count does not split the shape")
    g = np.random.default_rng(m + k)
    a = _r((g.standard_normal((m, k)) * 0.8 + 0.1).astype(np.float32))
    w = _r((g.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    b = g.standard_normal(n).astype(np.float32)
    r = _r((g.standard_normal((m, n)) * 1.3).astype(np.float32))
    A, W, B, R = _dev(a, dev), _dev(w, dev), torch.from_numpy(b).to(dev), _dev(r, dev)
    st = ops.new_row_stats(m, n, dev)
    st.fill_(float("nan"))
    out = ops.gemm_resid_split(A, W, B, R, stats_out=st)
    out_n = out.float().cpu().numpy()
    # float64 on sampled rows (the first / last panels, the seam between the full rounds and the tail, random rows)
    seam = (m // 256) * 256
    rows = np.unique(np.clip(np.concatenate([np.arange(0, 300), np.arange(m - 300, m), np.arange(21760 - 300, 21760 + 300),
                                             np.arange(seam - 40, seam + 40), g.integers(0, m, 500)]), 0, m - 1))
    ref = a[rows].astype(np.float64) @ w.astype(np.float64).T + b + r[rows]
    np.testing.assert_allclose(out_n[rows], ref, **TOL)
    # slab statistics = sums over the bf16 values stored, every row
    got = ops.reduce_row_stats(st, m, n).cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[:, 0], out_n.astype(np.float64).sum(1), rtol=1e-4, atol=3e-3)
    np.testing.assert_allclose(got[:, 1], (out_n.astype(np.float64) ** 2).sum(1), rtol=1e-4)
    # the unsplit launch: the same sums in another order -> equal up to one bf16 rounding step in a few elements
    base = ops.gemm(A, W, B, L.AG_EPI_BIAS_RESID, BF16, resid=R).float().cpu().numpy()
    diff = np.abs(base - out_n)
    assert (diff > 0).mean() < 0.02
    assert diff.max() <= 2.0 ** -7 * np.maximum(np.abs(base), 1.0).max()
    # deterministic
    again = ops.gemm_resid_split(A, W, B, R).float().cpu().numpy()
    np.testing.assert_array_equal(again, out_n)


def test_resid_split_refuses_shapes_that_do_not_split(cuda_device):
    from autognothi_amd import ops
    dev = cuda_device
    with torch.cuda.device(dev):
        assert ops.gemm_resid_split_scratch_bytes(302592, 768, 3072) == 0       # the benchmarked step: 13 rounds + a 90 % full one
        assert ops.gemm_resid_split_scratch_bytes(6304, 768, 768) == 0          # out-projection: too short a contraction
    a = torch.zeros((6304, 768), dtype=torch.bfloat16, device=dev)
    w = torch.zeros((768, 768), dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="does not split"):
        ops.gemm_resid_split(a, w, None, torch.zeros((6304, 768), dtype=torch.bfloat16, device=dev))


def test_encoder_small_batch_split_on_off(cuda_device, ag_knobs):
    """the full-depth ViT-base fixture (1 input x K = 32 masks = 6 304 token rows: fc2 of every layer splits) through the recipes'
    fw_* callables with the split on (default) and off: same outputs up to bf16 rounding, both within the bf16 bound of the reference."""
    from util import build_case, run_fixture_case
    from autognothi_amd import engine
    c = build_case("vit_base_l12")
    try:
        on = run_fixture_case(c, cuda_device, "bf16")
        ag_knobs(AG_GEMM_SPLIT=0)
        off = run_fixture_case(c, cuda_device, "bf16")
    finally:
        engine.set_precision("fp32")
    for k in ("v_s", "v_1"):
        np.testing.assert_allclose(on[k], off[k], rtol=0, atol=2e-2, err_msg=k)     # (two bf16 runs with different rounding decisions)
        np.testing.assert_allclose(on[k], c["g"][k], rtol=0, atol=2e-2, err_msg=k)
    assert np.abs(on["v_s"] - off["v_s"]).max() > 0      # (the split path did run: another summation order)


def test_last_layer_query_trim_on_off(cuda_device, ag_knobs):
    """cls_only_last: the last layer computes keys / values for every token and queries for the CLS rows only (AG_LAST_Q_TRIM, default) —
    against the untrimmed projection on the full-depth ViT-base fixture: the same surrogate outputs up to bf16 rounding of one query row per
    sequence (the CLS query goes through the LayerNorm kernel + the unfolded weights instead of the folded GEMM)."""
    from util import build_case, run_fixture_case
    from autognothi_amd import engine
    c = build_case("vit_base_l12")
    try:
        on = run_fixture_case(c, cuda_device, "bf16")
        ag_knobs(AG_LAST_Q_TRIM=0)
        off = run_fixture_case(c, cuda_device, "bf16")
    finally:
        engine.set_precision("fp32")
    for k in ("v_s", "v_1", "v_0"):
        np.testing.assert_allclose(on[k], off[k], rtol=0, atol=5e-3, err_msg=k)
        np.testing.assert_allclose(on[k], c["g"][k], rtol=0, atol=2e-2, err_msg=k)


@pytest.mark.parametrize("case", ["vit_base_l12", "vit_large_l24"])
def test_last_layer_without_kv_projection_on_off(cuda_device, ag_knobs, case):
    """cls_only_last: the last layer reads the CLS query only, so its keys and values are never projected (csrc/cls_last.hip:
    s_k = x_k . (W_k^T q) + b_k . q, o = W_v (sum_k p_k x_k) + b_v; AG_LAST_KV_SKIP, default) — against the projected form
    (AG_LAST_KV_SKIP=0: K / V GEMM + the CLS-only attention launch) on the reference-made full-depth fixtures: the same surrogate
    outputs up to bf16 rounding (the new form never rounds K and V), both within the bf16 bound of the reference."""
    from util import build_case, run_fixture_case
    from autognothi_amd import engine
    c = build_case(case)
    try:
        ag_knobs(AG_LAST_KV_SKIP=1)          # (the fixtures are one input x K masks: below the default row threshold of the path)
        on = run_fixture_case(c, cuda_device, "bf16")
        ag_knobs(AG_LAST_KV_SKIP=0)
        off = run_fixture_case(c, cuda_device, "bf16")
    finally:
        engine.set_precision("fp32")
    for k in ("v_s", "v_1", "v_0"):
        np.testing.assert_allclose(on[k], off[k], rtol=0, atol=5e-3, err_msg=k)
        np.testing.assert_allclose(on[k], c["g"][k], rtol=0, atol=2e-2, err_msg=k)
    assert np.abs(on["v_s"] - off["v_s"]).max() > 0      # (the path did run)
