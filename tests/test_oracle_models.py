"""Pin the oracle's masked ViT / BERT forward, explainer heads and loss against outputs of the reference
itself (tests/golden/model_*.npz, made by make_golden.py).  CPU only.  Tolerance: 1e-5 abs — the
reference's own coherency tolerance (scripts/train_all.py:212-215)."""
import numpy as np
import pytest

from oracle import shapley as osh
from oracle import torch_port as otp
from oracle import transformer as otr
from util import LTT_TAGS, MODEL_TAGS, build_case, state_dict_numpy

ATOL = 1e-5


def checks(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
    return np.concatenate([[a.sum(), np.abs(a).sum()], a[idx]])


@pytest.mark.parametrize("tag", MODEL_TAGS)
def test_oracle_matches_reference(tag):
    c = build_case(tag)
    g, prm, kind = c["g"], c["meta"]["params"], c["meta"]["kind"]
    sd_s, sd_e = state_dict_numpy(c["surrogate"]), state_dict_numpy(c["explainer"])
    xs_ext = np.repeat(c["xs"], c["K"], axis=0)
    ones1 = np.ones((1, c["P"]), dtype=np.int64)
    onesb = np.ones((c["B"], c["P"]), dtype=np.int64)
    srg = otr.vit_surrogate if kind == "vit" else otr.bert_surrogate
    exp = otr.vit_explainer if kind == "vit" else otr.bert_explainer

    trace = []
    v_s = srg(xs_ext, c["masks"], sd_s, prm, collect=trace)
    v_0 = srg(c["null"], ones1, sd_s, prm)
    v_1 = srg(c["xs"], onesb, sd_s, prm)
    np.testing.assert_allclose(v_s, g["v_s"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(v_0, g["v_0"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(v_1, g["v_1"], rtol=0, atol=ATOL)
    # the torch-CPU port used as bench.py's cpu_baseline must agree with the reference too
    import torch
    sd_t = {k: torch.from_numpy(v) for k, v in sd_s.items()}
    srg_t = otp.vit_surrogate if kind == "vit" else otp.bert_surrogate
    v_s_t = srg_t(torch.from_numpy(xs_ext), torch.from_numpy(c["masks"]), sd_t, prm).numpy()
    np.testing.assert_allclose(v_s_t, g["v_s"], rtol=0, atol=ATOL)
    # per-layer hidden-state checksums (layer outputs; trace[0] is the embedding output)
    for li in range(prm["num_hidden_layers"]):
        got, want = checks(trace[li + 1]), g["layer_trace"][li]
        np.testing.assert_allclose(got[2:], want[2:], rtol=0, atol=5e-5)
        np.testing.assert_allclose(got[:2], want[:2], rtol=1e-5)
    # all-zero / all-one mask rows are legal sampler outputs
    edge = np.stack([np.zeros(c["P"], dtype=np.int64), np.ones(c["P"], dtype=np.int64)])
    np.testing.assert_allclose(srg(np.repeat(c["xs"][:1], 2, axis=0), edge, sd_s, prm), g["v_edge"], rtol=0, atol=ATOL)

    out = exp(c["xs"], onesb, g["v_1"], g["v_0"], sd_e, prm, duo=c["meta"]["duo"])
    phi = out[0] if c["meta"]["duo"] else out
    np.testing.assert_allclose(phi, g["phi"], rtol=0, atol=2e-5)
    if c["meta"]["duo"]:
        np.testing.assert_allclose(out[1], g["exp_logits"], rtol=0, atol=2e-5)
    loss, dphi = osh.loss_shapley_new(c["B"], c["K"], c["P"], c["masks"], g["v_0"], g["v_s"], g["v_1"], g["phi"])
    np.testing.assert_allclose(loss, g["loss"][0], rtol=1e-5)
    np.testing.assert_allclose(dphi, g["dphi"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("tag", LTT_TAGS)
def test_oracle_ltt_matches_reference(tag):
    """LTT (ladder side network): surrogate (side + backbone heads), explainer, two-branch Final."""
    c = build_case(tag)
    g, prm, kind = c["g"], c["meta"]["params"], c["meta"]["kind"]
    sd_s, sd_e, sd_f = state_dict_numpy(c["surrogate"]), state_dict_numpy(c["explainer"]), state_dict_numpy(c["final"])
    xs_ext = np.repeat(c["xs"], c["K"], axis=0)
    ones1 = np.ones((1, c["P"]), dtype=np.int64)
    onesb = np.ones((c["B"], c["P"]), dtype=np.int64)
    srg = otr.ltt_vit_surrogate if kind == "vit" else otr.ltt_bert_surrogate
    exp = otr.ltt_vit_explainer if kind == "vit" else otr.ltt_bert_explainer
    fin = otr.ltt_vit_final if kind == "vit" else otr.ltt_bert_final
    trace = []
    v_s, v_s_cls = srg(xs_ext, c["masks"], sd_s, prm, collect=trace)
    np.testing.assert_allclose(v_s, g["v_s"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(v_s_cls, g["v_s_cls"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(srg(c["null"], ones1, sd_s, prm)[0], g["v_0"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(srg(c["xs"], onesb, sd_s, prm)[0], g["v_1"], rtol=0, atol=ATOL)
    for li in range(prm["num_hidden_layers"]):
        got, want = checks(trace[li + 1]), g["layer_trace"][li]
        np.testing.assert_allclose(got[2:], want[2:], rtol=0, atol=5e-5)
    edge = np.stack([np.zeros(c["P"], dtype=np.int64), np.ones(c["P"], dtype=np.int64)])
    np.testing.assert_allclose(srg(np.repeat(c["xs"][:1], 2, axis=0), edge, sd_s, prm)[0], g["v_edge"], rtol=0, atol=ATOL)
    phi, logits = exp(c["xs"], onesb, g["v_1"], g["v_0"], sd_e, prm)
    np.testing.assert_allclose(phi, g["phi"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(logits, g["exp_logits"], rtol=0, atol=ATOL)
    f_logits, f_phi = fin(c["xs"], sd_f, prm)
    np.testing.assert_allclose(f_logits, g["fin_logits"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(f_phi, g["fin_phi"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("tag", ["vit_base_l12", "bert_base_l12", "vit_large_l24"])
def test_oracle_full_depth_matches_reference(tag):
    """The shipped depth and K (BASELINE configs 2-4; fixtures: make_golden.py full_depth): the torch-CPU port — bench.py's
    cpu_baseline and the GPU tests' gradient reference — reproduces the reference's K-mask surrogate values at 12 / 24
    layers (the numpy oracle is pinned on the truncated fixtures above; at full depth it takes minutes on 8 cores)."""
    import torch
    c = build_case(tag)
    g, prm, kind = c["g"], c["meta"]["params"], c["meta"]["kind"]
    sd_t = {k: v.detach() for k, v in c["surrogate"].state_dict().items()}
    srg_t = otp.vit_surrogate if kind == "vit" else otp.bert_surrogate
    xs_ext = torch.from_numpy(np.repeat(c["xs"], c["K"], axis=0))
    v_s = srg_t(xs_ext, torch.from_numpy(c["masks"]), sd_t, prm).numpy()
    np.testing.assert_allclose(v_s, g["v_s"], rtol=0, atol=ATOL)
    v_1 = srg_t(torch.from_numpy(c["xs"]), torch.ones((c["B"], c["P"]), dtype=torch.long), sd_t, prm).numpy()
    np.testing.assert_allclose(v_1, g["v_1"], rtol=0, atol=ATOL)
