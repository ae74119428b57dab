"""Pin the oracle's Shapley reductions against reference outputs (CPU)."""
import numpy as np

from oracle import shapley as osh
from util import golden, unpack


def test_normalize_loss_kl():
    g = golden("shapley_fns.npz")
    for tag in ("vit", "bert"):
        b, k, p, c = [int(x) for x in g[f"{tag}_dims"]]
        out = osh.normalize_shapley_explanation(g[f"{tag}_norm_pred"], g[f"{tag}_norm_grand"], g[f"{tag}_norm_null"])
        np.testing.assert_allclose(out, g[f"{tag}_norm_out"], rtol=0, atol=1e-5)
        mask = unpack(g[f"{tag}_loss_mask"], p)
        loss, dphi = osh.loss_shapley_new(b, k, p, mask, g[f"{tag}_loss_v0"], g[f"{tag}_loss_vs"], g[f"{tag}_loss_v1"],
                                          g[f"{tag}_loss_phi"])
        np.testing.assert_allclose(loss, g[f"{tag}_loss_out"][0], rtol=1e-5)
        np.testing.assert_allclose(dphi, g[f"{tag}_loss_dphi"], rtol=1e-4, atol=1e-6)
        kl = osh.loss_logits_kl_divergence(g[f"{tag}_kl_ref"], g[f"{tag}_kl_cur"])
        np.testing.assert_allclose(kl, g[f"{tag}_kl_out"][0], rtol=1e-5, atol=1e-7)


def test_normalize_efficiency_gap_quirk():
    """Dividing by T = P+1 then dropping the CLS row means sum(phi) != v1 - v0 exactly."""
    g = golden("shapley_fns.npz")
    out = osh.normalize_shapley_explanation(g["vit_norm_pred"], g["vit_norm_grand"], g["vit_norm_null"])
    total_with_cls = out.sum(axis=1)
    np.testing.assert_allclose(total_with_cls, g["vit_norm_grand"] - g["vit_norm_null"], atol=2e-5)
    assert np.abs(out[:, 1:].sum(axis=1) - (g["vit_norm_grand"] - g["vit_norm_null"])).max() > 1e-4


def test_mc_permutation_shapley_matches_reference():
    """the oracle's permutation-Shapley loop vs the reference's own _get_shap output
    (tests/golden/mc_shapley.npz, 3 permutations on the 2-layer BERT fixture model)."""
    from oracle import transformer as otr
    from util import build_case, state_dict_numpy
    g = golden("mc_shapley.npz")
    c = build_case("bert_base_l2")
    sd, prm = state_dict_numpy(c["surrogate"]), c["meta"]["params"]
    ids = c["xs"][int(g["input_row"][0])][None]

    def probs(masks):
        return otr.bert_surrogate(np.repeat(ids, masks.shape[0], axis=0), masks, sd, prm)
    sv, v0, vn = osh.mc_permutation_shapley(probs, g["perms"])
    np.testing.assert_allclose(sv, g["sv"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(v0, g["v0"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(vn, g["vn"], rtol=0, atol=1e-5)


def test_perturbed_samples_with_ties_fixture():
    """The reference ranks attributions with np.argsort's default kind, which is not stable: inside a group of equal
    attributions the order is whatever the host's numpy build does (AVX-512 / AVX2 sorting networks or introsort).  The
    fixture (reference run on the build host) documents that: its ranking differs from the stable one, yet every mask is
    a valid top-k set; the oracle (same numpy call) is held to the same property, and to equality wherever the top-k set is
    unique."""
    from util import check_perturbed_against_reference
    g = golden("perturbed_ties.npz")
    assert any(not np.array_equal(g[f"c{i}_ranking_reference_host"], g[f"c{i}_ranking_stable"]) for i in range(len(g["cases"])))
    for i, (p, steps) in enumerate(g["cases"]):
        attr = g[f"c{i}_attr"]
        for base in (0, 1):
            stops, masks = osh.get_perturbed_samples(attr, int(p), int(steps), base)
            check_perturbed_against_reference(attr, base, stops, masks, g[f"c{i}_b{base}_stops"], unpack(g[f"c{i}_b{base}_masks"], int(p)))
