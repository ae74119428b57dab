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
