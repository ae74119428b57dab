"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unpack(bits, n):
    """inverse of np.packbits(..., axis=-1) for an n-wide last axis -> int64 0/1."""
    return np.unpackbits(bits, axis=-1)[..., :n].astype(np.int64)


def bits_to_mask(words, n_players):
    """uint32 key-bit words [R,Tw] (bit t = token t, token 0 = CLS) -> (cls column, [R,P] int64)."""
    w = np.asarray(words).astype(np.uint32)
    t = n_players + 1
    idx = np.arange(t)
    full = ((w[:, idx // 32] >> (idx % 32).astype(np.uint32)) & 1).astype(np.int64)
    return full[:, 0], full[:, 1:]


def state_dict_numpy(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


MODEL_TAGS = ["vit_tiny_c1", "vit_base_l2", "vit_large_l2", "bert_base_l2", "duo_bert_base_l2", "duo_vit_tiny_l3",
              "froyo_vit_tiny_l3", "froyo_bert_base_l2"]


LTT_TAGS = ["ltt_vit_tiny_l3", "ltt_bert_base_l2"]
# the shipped ladder (h = 96: 12 heads of 8, 384 intermediate) on the 12-layer base backbones, K = 32, one input
LTT_FULL_TAGS = ["ltt_bert_base_l12", "ltt_vit_base_l12"]

# the shipped configs at their real depth and K, one input each (BASELINE configs 2-5)
FULL_TAGS = ["vit_base_l12", "bert_base_l12", "vit_large_l24", "duo_bert_base_l12", "froyo_vit_base_l12"]


def recipe_kind(meta):
    if meta.get("ltt"):
        return "ltt_" + meta["kind"]
    pre = "duo_vanilla_" if meta["duo"] else ("froyo_" if meta["froyo"] else "vanilla_")
    return pre + meta["kind"]


def build_case(tag):
    """-> dict(meta, recipe, cfg, surrogate, explainer (synth weights, CPU), inputs Xs/null (numpy), masks [R,P])."""
    import torch
    from autognothi_amd.recipes import get_recipe
    from autognothi_amd.utils import synth
    meta = golden_json(f"model_{tag}.json")
    g = golden(f"model_{tag}.npz")
    recipe = get_recipe(recipe_kind(meta))
    cfg = recipe.t_config(**meta["params"])
    srg, exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
    synth.load_synth_weights(srg, seed=meta["weights"]["surrogate_seed"])
    synth.load_synth_weights(exp, seed=meta["weights"]["explainer_seed"])
    tweak = meta.get("explainer_head")      # make_golden.apply_head_tweak: scaled explainer parameters (bert_base_l12_phi)
    if tweak:
        with torch.no_grad():
            for name, prm_ in exp.named_parameters():
                for key, factor in tweak.get("scale", {}).items():
                    if key in name:
                        prm_.mul_(float(factor))
    srg.eval(); exp.eval()
    b, k, p = [int(x) for x in g["dims"]]
    prm = meta["params"]
    if meta["kind"] == "vit":
        xs = synth.synth_images(b, prm["img_px_size"], prm["img_channels"], seed=meta["input_seed"])
        null = np.zeros((1, prm["img_channels"], prm["img_px_size"], prm["img_px_size"]), dtype=np.float32)
    else:
        xs = synth.synth_token_ids(b, prm["max_position_embeddings"], prm["vocab_size"], seed=meta["input_seed"])
        null = synth.synth_null_ids(prm["max_position_embeddings"], prm["vocab_size"])
    out = dict(meta=meta, g=g, recipe=recipe, cfg=cfg, surrogate=srg, explainer=exp, xs=xs, null=null,
               masks=unpack(g["masks"], p), B=b, K=k, P=p)
    if "final_seed" in meta["weights"]:
        fin = recipe.t_final(cfg)
        synth.load_synth_weights(fin, seed=meta["weights"]["final_seed"])
        out["final"] = fin.eval()
    return out


def bf16_deviation(got, g, ref):
    """the figures bench.py reports as secondary.bf16_vs_reference: deviation of a bf16 run from the fp32 reference fixture,
    next to the reference's own autocast(bf16) deviation (ref = model_<tag>_bf16ref.npz)."""
    e_vs, e_phi = np.abs(got["v_s"] - g["v_s"]), np.abs(got["phi"] - g["phi"])
    phi_max = float(np.abs(g["phi"]).max())
    return dict(v_s_max_abs=float(e_vs.max()), v_s_rms=float(np.sqrt((e_vs ** 2).mean())),
                v_1_max_abs=float(np.abs(got["v_1"] - g["v_1"]).max()),
                phi_max_abs=float(e_phi.max()), phi_max_abs_over_max_phi=float(e_phi.max()) / phi_max,
                phi_rms_over_max_phi=float(np.sqrt((e_phi ** 2).mean())) / phi_max,
                reference_autocast_bf16=dict(v_s_max_abs=float(ref["v_s_maxabs"][0]), v_s_rms=float(ref["v_s_rms"][0]),
                                             phi_max_abs_over_max_phi=float(ref["phi_maxabs"][0]) / phi_max))


def run_fixture_case(c, dev, precision, share_inputs=True):
    """the fixture's calls through the recipes' fw_* callables on the HIP path (same sequence as make_golden.gen_model_fixture) -> numpy dict."""
    import torch
    from autognothi_amd import engine, ops
    engine.set_precision(precision)
    recipe = c["recipe"]
    srg, exp = c["surrogate"].to(dev), c["explainer"].to(dev)
    xs = torch.from_numpy(c["xs"]).to(dev)
    null = torch.from_numpy(c["null"]).to(dev)
    masks = torch.from_numpy(c["masks"]).to(dev)
    ones1 = torch.ones((1, c["P"]), dtype=torch.long, device=dev)
    onesb = torch.ones((c["B"], c["P"]), dtype=torch.long, device=dev)
    with torch.no_grad():
        v_0, _ = recipe.fw_surrogate(srg, null, ones1)
        xin = xs if share_inputs else torch.repeat_interleave(xs, c["K"], dim=0)
        v_s, _ = recipe.fw_surrogate(srg, xin, masks)
        v_1, _ = recipe.fw_surrogate(srg, xs, onesb)
        edge = torch.stack([torch.zeros(c["P"], dtype=torch.long), torch.ones(c["P"], dtype=torch.long)]).to(dev)
        v_edge, _ = recipe.fw_surrogate(srg, xs[:1], edge)
        g = c["g"]
        phi, extra = recipe.fw_explainer(exp, xs, onesb, torch.from_numpy(g["v_1"]).to(dev), torch.from_numpy(g["v_0"]).to(dev))
        bits = ops.pack_mask(masks)
        loss, dphi = ops.shapley_loss(bits, torch.from_numpy(g["v_0"]).to(dev), torch.from_numpy(g["v_s"]).to(dev),
                                      torch.from_numpy(g["phi"]).to(dev), c["B"], c["K"])
    out = dict(v_0=v_0, v_s=v_s, v_1=v_1, v_edge=v_edge, phi=phi, loss=loss, dphi=dphi)
    if extra is not None:
        out["exp_logits"] = extra
    return {k: v.float().cpu().numpy() for k, v in out.items()}


def tie_split(attr, stop):
    """does a cut after the `stop` highest attributions fall INSIDE a group of equal values (top-`stop` set not unique)?"""
    v = np.sort(np.asarray(attr))[::-1]
    return 0 < stop < len(v) and v[stop - 1] == v[stop]


def check_perturbed_against_reference(attr, base, stops, masks, ref_stops, ref_masks):
    """masks [S,P] vs the reference's on an attribution with ties: identical wherever the top-k set is unique; where a cut
    splits a tie group any k-subset that contains everything strictly larger and nothing strictly smaller is a correct
    answer (the reference's own choice there is np.argsort's unstable, host-specific order: tests/golden/perturbed_ties.json)."""
    attr = np.asarray(attr)
    assert np.array_equal(stops, ref_stops)
    n_split = 0
    for s, st in enumerate(stops):
        flipped = masks[s] != base
        assert flipped.sum() == st
        if not tie_split(attr, st):
            assert np.array_equal(masks[s], ref_masks[s]), (s, st)
            continue
        n_split += 1
        thr = np.sort(attr)[::-1][st - 1]
        assert flipped[attr > thr].all() and not flipped[attr < thr].any(), (s, st)
        ref_flipped = ref_masks[s] != base
        assert ref_flipped[attr > thr].all() and not ref_flipped[attr < thr].any()
    return n_split
