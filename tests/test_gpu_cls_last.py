"""GPU parity of csrc/cls_last.hip — the last layer's attention of a CLS-only ViT forward WITHOUT its key / value projection — at the
kernel level, against a float64 numpy restatement of the PROJECTED form it replaces (reference models/vanilla_vit.py:436-465 for the
one query the heads read, models/vanilla_vit.py:51-56; LayerNorm models/vanilla_vit.py:369):

    x = LayerNorm(h);  K = x W_k^T + b_k;  V = x W_v^T + b_v;  s = (q . K_t) / 8;  s *= mask;  p = softmax(s);  ctx = p V   (per head)

on random operands (the same bf16-rounded folded weights, queries and residual rows the kernel reads), R in {64, 200, 1 536} rows,
H in {768, 1 024}, rows with every player masked and with every player visible included; and, through the whole pipeline, against the
reference-made two-input fixture whose 64 masked rows take the path at the DEFAULT threshold (AG_LAST_KV_SKIP = 64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = 197
EPS = 1e-12


def _operands(r, h, seed, dev):
    g = torch.Generator().manual_seed(seed)
    heads = h // 64
    hid = (torch.randn((r * T, h), generator=g) * 1.3 + 0.2 * torch.randn((1, h), generator=g)).to(torch.bfloat16)
    gamma = 1.0 + 0.2 * torch.randn(h, generator=g)
    beta = 0.1 * torch.randn(h, generator=g)
    w_kv = torch.randn((2 * h, h), generator=g) / h ** 0.5
    b_kv = 0.1 * torch.randn(2 * h, generator=g)
    w_kv_ln = (w_kv * gamma[None, :]).to(torch.bfloat16)              # gamma (.) W, rounded (what engine.PackedFoldedLinear packs)
    b_kv_ln = (b_kv.double() + w_kv.double() @ beta.double()).float()  # b + W beta
    q = (torch.randn((r, h), generator=g) * 0.8).to(torch.bfloat16)   # the CLS queries, already projected
    masks = (torch.rand((r, T - 1), generator=g) > torch.rand((r, 1), generator=g)).long()
    masks[0] = 0          # every player masked: only the CLS key is visible
    masks[1] = 1          # every player visible
    masks[-1] = 0
    return heads, hid, w_kv_ln, b_kv_ln, q, masks


def _projected_reference(hid, w_kv_ln, b_kv_ln, q, masks, rows, h, heads):
    """float64, rows `rows` only (rows are independent).  The folded form IS LayerNorm + Linear: x_hat = (h - mean) rstd, K = x_hat
    (gamma (.) W_k)^T + (b_k + W_k beta) (DESIGN §3 'LayerNorm folding')."""
    hd = hid.double().numpy().reshape(-1, T, h)
    wk, wv = w_kv_ln[:h].double().numpy(), w_kv_ln[h:].double().numpy()
    bk, bv = b_kv_ln[:h].double().numpy(), b_kv_ln[h:].double().numpy()
    out = np.zeros((len(rows), h))
    for i, r in enumerate(rows):
        x = hd[r]
        mean = x.mean(axis=1, keepdims=True)
        var = (x * x).mean(axis=1, keepdims=True) - mean * mean
        xh = (x - mean) / np.sqrt(np.maximum(var, 0.0) + EPS)
        k = xh @ wk.T + bk
        v = xh @ wv.T + bv
        m = np.concatenate([[1.0], masks[r].double().numpy()])      # CLS key always visible (recipes/vanilla_vit.py:219-224)
        qr = q[r].double().numpy()
        for a in range(heads):
            sl = slice(64 * a, 64 * a + 64)
            s = (k[:, sl] @ qr[sl]) / 8.0
            s = s * m                                               # ViT: scores * mask (models/vanilla_vit.py:449-450): a masked key keeps logit 0
            p = np.exp(s - s.max())
            p /= p.sum()
            out[i, sl] = p @ v[:, sl]
    return out


@pytest.mark.parametrize("r,h", [(64, 768), (200, 768), (1536, 768), (64, 1024), (200, 1024), (1536, 1024)])
def test_cls_last_attention_vs_float64_projected_form(cuda_device, r, h):
    from autognothi_amd import _lib as L, ops
    dev = cuda_device
    heads, hid, w_kv_ln, b_kv_ln, q, masks = _operands(r, h, 1000 + r + h, dev)
    assert L.lib().ag_cls_last_is_supported(T, h, heads, L.AG_BF16) == 1
    d_h, d_w, d_b, d_q = hid.to(dev), w_kv_ln.to(dev), b_kv_ln.to(dev), q.to(dev)
    bits = ops.pack_mask(masks.to(dev))
    stats = ops.row_stats(d_h)                                       # 256-column slabs, as the producing GEMM's epilogue writes them
    ctx = torch.zeros((r, h), dtype=torch.bfloat16, device=dev)
    nbytes = int(L.lib().ag_cls_last_workspace_bytes(r, h, heads))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with L.on(dev):
        L.check(L.lib().ag_cls_last_attention_rows(L.ptr(d_h), L.ptr(stats), 256, L.ptr(bits), L.ptr(d_q), L.ptr(d_w), L.ptr(d_b), EPS,
                                                   L.ptr(ctx), h, r, T, h, heads, L.ptr(scratch), nbytes, L.stream()))
    torch.cuda.synchronize()
    got = ctx.float().cpu().numpy()
    rows = sorted(set([0, 1, 2, 3, r // 2, r // 2 + 1, r - 2, r - 1] + list(range(7, r, max(1, r // 24)))))
    want = _projected_reference(hid, w_kv_ln, b_kv_ln, q, masks, rows, h, heads)
    scale = float(np.abs(want).max())
    # rtol 1e-2 on the context rows + the bf16 rounding of an output at the rows' magnitude (the kernel rounds Wt = W_k^T q, zhat and
    # the output once each to bf16; the projected form would have rounded K and V instead)
    np.testing.assert_allclose(got[rows], want, rtol=1e-2, atol=6e-3 * scale)
    err = np.abs(got[rows] - want)
    print(f"cls_last R={r} H={h}: max abs err {err.max():.3e} (scale {scale:.3f}), rms {np.sqrt((err ** 2).mean()):.3e}")
    # the all-masked rows: the soft-max over 197 logits of which 196 are exactly 0 (not -inf): every value row still counts
    assert np.isfinite(got).all()


def test_two_input_fixture_takes_the_path_by_default(cuda_device, ag_knobs):
    """reference-made fixture: ViT-base, 12 layers, 2 inputs x K = 32 = 64 masked rows (tests/golden/make_golden.py full_depth_b2): the
    encoder's DEFAULT row threshold sends its last layer through cls_last.hip; outputs within the bf16 bound of the reference's own
    autocast run (tests/test_gpu_fulldepth.BF16_BOUNDS["vit_base_l12"]) and different from the projected form's (the path did run)."""
    from util import build_case, run_fixture_case
    from autognothi_amd import engine
    from test_gpu_fulldepth import BF16_BOUNDS
    c = build_case("vit_base_l12_b2")
    assert c["g"]["v_s"].shape[0] == 64
    try:
        on = run_fixture_case(c, cuda_device, "bf16")               # no knob: the default threshold
        ag_knobs(AG_LAST_KV_SKIP=0)
        off = run_fixture_case(c, cuda_device, "bf16")
    finally:
        engine.set_precision("fp32")
    vmax, vrms, phimax = BF16_BOUNDS["vit_base_l12"]
    for got in (on, off):
        d = got["v_s"] - c["g"]["v_s"]
        assert float(np.abs(d).max()) <= vmax and float(np.sqrt((d ** 2).mean())) <= vrms
        assert float(np.abs(got["phi"] - c["g"]["phi"]).max()) <= phimax * float(np.abs(c["g"]["phi"]).max())
    assert np.abs(on["v_s"] - off["v_s"]).max() > 0
