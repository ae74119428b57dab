#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the *reference itself*.

Run ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's Python (models/shapley.py, recipes/*, scripts/measure_faithfulness.py
helper) unmodified, feeds it deterministic inputs / name-keyed synthetic weights
(autognothi_amd/utils/synth.py) and stores inputs that cannot be regenerated plus the expected
outputs as small .npz/.json files.  Fixtures are data only; no reference source is stored.
"""
import importlib.machinery
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root")

from autognothi_amd.utils import synth  # noqa: E402

import reference.models.shapley as rshap  # noqa: E402
from reference.utils.tools import set_iterative_seed  # noqa: E402
import reference.recipes.vanilla_vit as r_vvit  # noqa: E402
import reference.recipes.vanilla_bert as r_vbert  # noqa: E402
import reference.recipes.duo_vanilla_vit as r_dvit  # noqa: E402
import reference.recipes.duo_vanilla_bert as r_dbert  # noqa: E402
import reference.recipes.froyo_vit as r_fvit  # noqa: E402
import reference.recipes.froyo_bert as r_fbert  # noqa: E402
import reference.recipes.ltt_vit as r_lvit  # noqa: E402
import reference.recipes.ltt_bert as r_lbert  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def hparams(exp):
    with open(f"/root/reference/experiments/{exp}/.hparams.json") as f:
        return json.load(f)["net"]["params"]


# ------------------------------------------------------------------ seeds
def gen_seeds():
    import hashlib
    out = {}
    for key in ["train_explainer[epoch=1]", "train_explainer[epoch=2]", "train_explainer[epoch=3]",
                "train_surrogate[epoch=1]", "stage-a"]:
        set_iterative_seed(3407, key)
        out[key] = int(torch.initial_seed())
    with open(os.path.join(HERE, "iterative_seeds.json"), "w") as f:
        json.dump({"master": 3407, "derived": out}, f, indent=1)
    print("wrote iterative_seeds.json", out)
    return out


# ------------------------------------------------------------------ masks
def gen_masks(seeds):
    arrs = {}
    tables = {}
    for P in (196, 127, 511):
        probs = torch.arange(1, P) * (P - torch.arange(1, P))
        probs = 1 / probs
        probs = probs / probs.sum()
        prefix = torch.cumsum(probs, dim=0) - probs
        tables[f"prefix_{P}"] = prefix.numpy()
        tables[f"probs_{P}"] = probs.numpy()
    save("prefix_tables.npz", **tables)

    cases = []
    seed_list = [0, 3407, seeds["train_explainer[epoch=1]"], seeds["train_explainer[epoch=2]"]]
    for s in seed_list:
        for (R, P) in [(8, 196), (32, 196), (64, 196), (32, 127), (4, 511), (2, 196)]:
            torch.manual_seed(s)
            m1 = rshap.mask_shapley_new(R, P)
            m2 = rshap.mask_shapley_new(R, P)  # stream continuity across calls
            key = f"s{s}_R{R}_P{P}"
            arrs[key + "_a"] = np.packbits(m1.numpy().astype(np.uint8), axis=1)
            arrs[key + "_b"] = np.packbits(m2.numpy().astype(np.uint8), axis=1)
            cases.append([int(s), R, P])
    arrs["cases"] = np.asarray(cases, dtype=np.int64)
    # a large draw that crosses many twists
    torch.manual_seed(99)
    big = rshap.mask_shapley_new(1024, 196)
    arrs["big_s99_R1024_P196_rowsum"] = big.sum(1).numpy()
    arrs["big_s99_R1024_P196_colsum"] = big.sum(0).numpy()
    import zlib
    arrs["big_s99_R1024_P196_crc"] = np.asarray([zlib.crc32(np.packbits(big.numpy().astype(np.uint8), axis=1).tobytes())], dtype=np.int64)
    save("masks_shapley.npz", **arrs)

    arrs = {}
    cases = []
    for s in (0, 3407):
        for (B, P) in [(8, 196), (8, 127), (3, 511)]:
            torch.manual_seed(s)
            m = rshap.mask_purely_uniform(B, P)
            arrs[f"uniform_s{s}_B{B}_P{P}"] = np.packbits(m.numpy().astype(np.uint8), axis=1)
            random.seed(s)
            m = rshap.mask_uniform_selective(B, P, P // 3)
            arrs[f"selective_s{s}_B{B}_P{P}"] = np.packbits(m.numpy().astype(np.uint8), axis=1)
            cases.append([s, B, P])
    arrs["cases"] = np.asarray(cases, dtype=np.int64)
    save("masks_other.npz", **arrs)


# ------------------------------------------------------------------ shapley fns
def gen_shapley_fns():
    g = np.random.default_rng(1234)
    arrs = {}
    for tag, (B, K, P, C) in {"vit": (3, 8, 196, 10), "bert": (2, 6, 127, 2)}.items():
        T = P + 1
        pred = g.standard_normal((B, T, C)).astype(np.float32)
        grand = g.random((B, C)).astype(np.float32)
        null = g.random((1, C)).astype(np.float32)
        out = rshap.normalize_shapley_explanation(torch.from_numpy(pred), torch.from_numpy(grand), torch.from_numpy(null))
        arrs[f"{tag}_norm_pred"], arrs[f"{tag}_norm_grand"], arrs[f"{tag}_norm_null"] = pred, grand, null
        arrs[f"{tag}_norm_out"] = out.numpy()

        torch.manual_seed(11)
        mask = rshap.mask_shapley_new(B * K, P).reshape(B, K, P)
        v_s = g.random((B * K, C)).astype(np.float32)
        v_1 = g.random((B, C)).astype(np.float32)
        phi = (0.05 * g.standard_normal((B, C, P))).astype(np.float32)
        phi_t = torch.from_numpy(phi).requires_grad_(True)
        loss = rshap.loss_shapley_new(B, K, P, mask, torch.from_numpy(null), torch.from_numpy(v_s), torch.from_numpy(v_1), phi_t)
        loss.backward()
        arrs[f"{tag}_loss_mask"] = np.packbits(mask.numpy().astype(np.uint8), axis=2)
        arrs[f"{tag}_loss_v0"], arrs[f"{tag}_loss_vs"], arrs[f"{tag}_loss_v1"], arrs[f"{tag}_loss_phi"] = null, v_s, v_1, phi
        arrs[f"{tag}_loss_out"] = np.asarray([loss.item()], dtype=np.float32)
        arrs[f"{tag}_loss_dphi"] = phi_t.grad.numpy()
        arrs[f"{tag}_dims"] = np.asarray([B, K, P, C], dtype=np.int64)

        ref = torch.softmax(torch.from_numpy(g.standard_normal((5, C)).astype(np.float32)), -1)
        cur = torch.softmax(torch.from_numpy(g.standard_normal((5, C)).astype(np.float32)), -1)
        kl = rshap.loss_logits_kl_divergence(ref, cur)
        arrs[f"{tag}_kl_ref"], arrs[f"{tag}_kl_cur"] = ref.numpy(), cur.numpy()
        arrs[f"{tag}_kl_out"] = np.asarray([kl.item()], dtype=np.float32)
    save("shapley_fns.npz", **arrs)


# ------------------------------------------------------------------ faithfulness masks
def _stub_modules():
    import transformers, datasets  # noqa: F401  (must be imported before the stand-ins)
    def mk(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    class _Any:
        def __init__(self, *a, **k): pass
    tv = mk("torchvision")
    tvt = mk("torchvision.transforms", **{n: _Any for n in
              ["CenterCrop", "ColorJitter", "Compose", "Lambda", "Normalize", "RandomHorizontalFlip", "RandomResizedCrop",
               "RandomVerticalFlip", "Resize", "ToTensor"]})
    tvf = mk("torchvision.transforms.functional", resize=lambda *a, **k: None)
    tv.transforms = tvt
    tvt.functional = tvf
    mk("wandb", Image=_Any, log=lambda *a, **k: None, init=lambda *a, **k: None, run=None)
    mk("shap", KernelExplainer=_Any, kmeans=lambda *a, **k: None)


def gen_perturbed():
    _stub_modules()
    from reference.scripts.measure_faithfulness import _get_perturbed_samples, _auc
    g = np.random.default_rng(77)
    arrs = {}
    cases = []
    for i, (P, steps) in enumerate([(196, 4), (196, 32), (196, 300), (127, 8), (127, 127), (511, 16)]):
        attr = g.standard_normal(P).astype(np.float32)
        for base in (0, 1):
            stops, masks = _get_perturbed_samples(torch.from_numpy(attr), P, steps, base)
            arrs[f"c{i}_b{base}_stops"] = stops.numpy()
            arrs[f"c{i}_b{base}_masks"] = np.packbits(masks.numpy().astype(np.uint8), axis=1)
        arrs[f"c{i}_attr"] = attr
        cases.append([P, steps])
    arrs["cases"] = np.asarray(cases, dtype=np.int64)
    curve = {int(k): float(v) for k, v in zip(range(0, 50, 5), g.random(10))}
    arrs["auc_vals"] = np.asarray(list(curve.values()), dtype=np.float64)
    arrs["auc_out"] = np.asarray([_auc(curve)], dtype=np.float64)
    save("perturbed.npz", **arrs)


# ------------------------------------------------------------------ model fixtures
def checks(t):
    a = t.detach().double().reshape(-1)
    idx = np.linspace(0, a.numel() - 1, 16).astype(np.int64)
    return np.asarray([a.sum().item(), a.abs().sum().item()] + a[idx].tolist(), dtype=np.float64)


def layer_trace(model_backbone_layers, run):
    """Run `run()` with forward hooks on each encoder layer; returns per-layer checksums."""
    outs = []
    hooks = [ly.register_forward_hook(lambda m, i, o: outs.append(checks(o))) for ly in model_backbone_layers]
    try:
        res = run()
    finally:
        for h in hooks:
            h.remove()
    return res, np.stack(outs)


def apply_head_tweak(m_exp, tweak):
    """`tweak` = {"scale": {name substring: factor}}: scales the matching explainer parameters after the synthetic load (the same
    tweak is applied by tests/util.build_case on the HIP side).  Round 4 uses it for bert_base_l12_phi: a random-weight post-LN
    BERT averages its tokens towards one another layer after layer, so the explainer head sees nearly the same vector at every
    token and phi = pred - mean_t(pred) + ... is a small difference of large numbers (bert_base_l12: max|pred| 0.174, max|phi| 0.017);
    with the attention output projections scaled by 0.3 the tokens stay distinct, as in a trained model, and max|phi| ~ max|pred|."""
    if not tweak:
        return
    with torch.no_grad():
        for name, p in m_exp.named_parameters():
            for key, factor in tweak.get("scale", {}).items():
                if key in name:
                    p.mul_(float(factor))


def gen_model_fixture(tag, recipe_fn, params, kind, B, K, mask_seed, duo=False, froyo=False, ltt=False, head=None):
    recipe = recipe_fn()
    cfg = recipe.t_config(**params)
    P = recipe.n_players(cfg)
    m_srg = recipe.t_surrogate(cfg)
    m_exp = recipe.t_explainer(cfg)
    synth.load_synth_weights(m_srg, seed=0)
    synth.load_synth_weights(m_exp, seed=1)
    apply_head_tweak(m_exp, head)
    m_srg.eval(); m_exp.eval()
    if kind == "vit":
        Xs = torch.from_numpy(synth.synth_images(B, params["img_px_size"], params["img_channels"], seed=0))
        null = torch.zeros((1, params["img_channels"], params["img_px_size"], params["img_px_size"]))
        backbone_layers = m_srg.vit.encoder.layers
    else:
        L = params["max_position_embeddings"]
        Xs = torch.from_numpy(synth.synth_token_ids(B, L, params["vocab_size"], seed=0))
        null = torch.from_numpy(synth.synth_null_ids(L, params["vocab_size"]))
        backbone_layers = m_srg.bert.encoder.layers
    torch.manual_seed(mask_seed)
    masks = rshap.mask_shapley_new(B * K, P)
    Xs_ext = torch.repeat_interleave(Xs, K, dim=0)
    ones = torch.ones((B, P), dtype=torch.long)
    with torch.no_grad():
        v_0, _ = recipe.fw_surrogate(m_srg, null, torch.ones((1, P), dtype=torch.long))
        (v_s, _), trace = layer_trace(backbone_layers, lambda: recipe.fw_surrogate(m_srg, Xs_ext, masks))
        v_1, _ = recipe.fw_surrogate(m_srg, Xs, ones)
        # an all-zero mask row and an all-one row are legal sampler outputs: pin them too
        edge_masks = torch.stack([torch.zeros(P, dtype=torch.long), torch.ones(P, dtype=torch.long)])
        v_edge, _ = recipe.fw_surrogate(m_srg, Xs[:1].repeat_interleave(2, 0), edge_masks)
    aux = {}
    if ltt:   # side-network recipes: the second output of fw_surrogate is the frozen backbone's own prediction, and
        # Final runs two ladders off one backbone pass: pin both
        with torch.no_grad():
            _, v_s_cls = recipe.fw_surrogate(m_srg, Xs_ext, masks)
            m_fin = recipe.t_final(cfg)
            synth.load_synth_weights(m_fin, seed=2)
            m_fin.eval()
            fin_logits, fin_phi = recipe.fw_final(m_fin, Xs)
        aux = dict(v_s_cls=v_s_cls.numpy(), fin_logits=fin_logits.numpy(), fin_phi=fin_phi.numpy())
    phi, extra = recipe.fw_explainer(m_exp, Xs, ones, v_1, v_0)
    phi_leaf = phi.detach().clone().requires_grad_(True)
    loss = rshap.loss_shapley_new(B, K, P, masks.reshape(B, K, P), v_0, v_s, v_1, phi_leaf)
    loss.backward()
    arrs = dict(
        dims=np.asarray([B, K, P], dtype=np.int64), mask_seed=np.asarray([mask_seed], dtype=np.int64),
        masks=np.packbits(masks.numpy().astype(np.uint8), axis=1),
        v_0=v_0.numpy(), v_s=v_s.numpy(), v_1=v_1.numpy(), v_edge=v_edge.numpy(),
        phi=phi.detach().numpy(), loss=np.asarray([loss.item()], dtype=np.float32), dphi=phi_leaf.grad.numpy(),
        layer_trace=trace,
    )
    arrs.update(aux)
    if extra is not None:
        arrs["exp_logits"] = extra.detach().numpy()
    with open(os.path.join(HERE, f"model_{tag}.json"), "w") as f:
        meta = {"kind": kind, "duo": duo, "froyo": froyo, "params": params, "B": B, "K": K,
                "weights": {"surrogate_seed": 0, "explainer_seed": 1}, "input_seed": 0}
        if ltt:
            meta["ltt"] = True
            meta["weights"]["final_seed"] = 2
        if head:
            meta["explainer_head"] = head
        json.dump(meta, f, indent=1)
    save(f"model_{tag}.npz", **arrs)


def gen_models():
    tiny = hparams("vit_tiny_imagenette_vanilla")
    gen_model_fixture("vit_tiny_c1", r_vvit.vanilla_vit_recipe, tiny, "vit", B=2, K=4, mask_seed=3407)
    base = dict(hparams("vit_base_imagenette_vanilla"), num_hidden_layers=2)
    gen_model_fixture("vit_base_l2", r_vvit.vanilla_vit_recipe, base, "vit", B=1, K=4, mask_seed=3407)
    large = dict(hparams("vit_large_imagenette_vanilla"), num_hidden_layers=2)
    gen_model_fixture("vit_large_l2", r_vvit.vanilla_vit_recipe, large, "vit", B=1, K=2, mask_seed=3407)
    bert = dict(hparams("bert_base_tayp_vanilla"), num_hidden_layers=2, max_position_embeddings=128)
    gen_model_fixture("bert_base_l2", r_vbert.vanilla_bert_recipe, bert, "bert", B=2, K=4, mask_seed=3407)
    dbert = dict(hparams("bert_base_tayp_duo_vanilla"), num_hidden_layers=2, max_position_embeddings=128)
    gen_model_fixture("duo_bert_base_l2", r_dbert.duo_vanilla_bert_recipe, dbert, "bert", B=2, K=4, mask_seed=3407, duo=True)
    dvit = dict(tiny, num_hidden_layers=3)
    gen_model_fixture("duo_vit_tiny_l3", r_dvit.duo_vanilla_vit_recipe, dvit, "vit", B=2, K=4, mask_seed=3407, duo=True)
    gen_model_fixture("froyo_vit_tiny_l3", r_fvit.froyo_vit_recipe, dvit, "vit", B=2, K=4, mask_seed=3407, froyo=True)
    fbert = dict(hparams("bert_base_tayp_froyo"), num_hidden_layers=2, max_position_embeddings=128)
    gen_model_fixture("froyo_bert_base_l2", r_fbert.froyo_bert_recipe, fbert, "bert", B=2, K=4, mask_seed=3407, froyo=True)


def gen_ltt_models():
    """LTT (ladder side network): a 3-layer ViT-tiny backbone with a 24-wide ladder (3 heads of 8, the
    shipped head width) and the shipped BERT LTT config truncated to 2 layers (96-wide ladder, 12 heads of 8)."""
    tiny = hparams("vit_tiny_imagenette_vanilla")
    lvit = {k: v for k, v in tiny.items() if not k.startswith("explainer_")}
    lvit.update(num_hidden_layers=3, explainer_s_attn_num_layers=1, explainer_s_head_hidden_size=128,
                explainer_normalize=True, s_attn_hidden_size=24, s_attn_intermediate_size=96)
    gen_model_fixture("ltt_vit_tiny_l3", r_lvit.ltt_vit_recipe, lvit, "vit", B=2, K=4, mask_seed=3407, ltt=True)
    lbert = dict(hparams("bert_base_tayp_ltt"), num_hidden_layers=2, max_position_embeddings=128, explainer_s_head_hidden_size=256)
    gen_model_fixture("ltt_bert_base_l2", r_lbert.ltt_bert_recipe, lbert, "bert", B=2, K=4, mask_seed=3407, ltt=True)


def gen_mc_shapley():
    """Monte-Carlo permutation Shapley: the reference's own _get_shap (scripts/preview_text_shapley.py:62-132)
    on the 2-layer BERT fixture model; the permutations it drew are recorded (torch.randperm on the host generator)."""
    _stub_modules()
    import reference.scripts.preview_text_shapley as pts
    meta = json.load(open(os.path.join(HERE, "model_bert_base_l2.json")))
    params = meta["params"]
    recipe = r_vbert.vanilla_bert_recipe()
    cfg = recipe.t_config(**params)
    m_srg = recipe.t_surrogate(cfg)
    synth.load_synth_weights(m_srg, seed=0)
    m_srg.eval()
    L = params["max_position_embeddings"]
    ids = torch.from_numpy(synth.synth_token_ids(2, L, params["vocab_size"], seed=0))[1:2]
    P, reps = L - 1, 3
    torch.manual_seed(1234)
    perms = torch.stack([torch.randperm(P) for _ in range(reps)])
    torch.manual_seed(1234)
    sv, v0, vn = pts._get_shap(device=torch.device("cpu"), m_recipe=recipe, m_surrogate=m_srg,
                               gen_input=lambda a, b: (ids, torch.zeros(1, dtype=torch.long)), n_players=P, _inputs="x",
                               reps=reps, batch_size=16)
    save("mc_shapley.npz", perms=perms.numpy(), sv=sv.numpy(), v0=v0.numpy(), vn=vn.numpy(), input_row=np.asarray([1], dtype=np.int64),
         reps=np.asarray([reps], dtype=np.int64))


def gen_ltt_state_keys():
    tiny = hparams("vit_tiny_imagenette_vanilla")
    lvit = {k: v for k, v in tiny.items() if not k.startswith("explainer_")}
    lvit.update(num_hidden_layers=2, explainer_s_attn_num_layers=1, explainer_s_head_hidden_size=128,
                explainer_normalize=True, s_attn_hidden_size=24, s_attn_intermediate_size=96)
    lbert = dict(hparams("bert_base_tayp_ltt"), num_hidden_layers=1, max_position_embeddings=128)
    out = {}
    for kind, (fn, params) in {"ltt_vit": (r_lvit.ltt_vit_recipe, lvit), "ltt_bert": (r_lbert.ltt_bert_recipe, lbert)}.items():
        rec = fn()
        cfg = rec.t_config(**params)
        out[kind] = {"params": params, "roles": {
            role: {k: list(v.shape) for k, v in getattr(rec, "t_" + role)(cfg).state_dict().items()}
            for role in ("classifier", "surrogate", "explainer", "final")}}
    with open(os.path.join(HERE, "state_keys_ltt.json"), "w") as f:
        json.dump(out, f)
    print("wrote state_keys_ltt.json")


def gen_full_depth():
    """Full-depth fixtures: the shipped configs at their real depth and K, one input each, so the
    benchmarked kernels (ring GEMM at M >= 1024, LayerNorm fold, full 12/24-layer drift) meet the reference's numbers:
    ViT-base 12 layers K=32 (BASELINE config 2), BERT-base 12 layers L=128 K=32 (config 3), ViT-large 24 layers K=64
    (config 4), and config 5's two recipes (duo BERT-base, froyo ViT-base) at K=32."""
    base = hparams("vit_base_imagenette_vanilla")
    gen_model_fixture("vit_base_l12", r_vvit.vanilla_vit_recipe, base, "vit", B=1, K=32, mask_seed=3407)
    bert = dict(hparams("bert_base_tayp_vanilla"), max_position_embeddings=128)
    gen_model_fixture("bert_base_l12", r_vbert.vanilla_bert_recipe, bert, "bert", B=1, K=32, mask_seed=3407)
    large = hparams("vit_large_imagenette_vanilla")
    gen_model_fixture("vit_large_l24", r_vvit.vanilla_vit_recipe, large, "vit", B=1, K=64, mask_seed=3407)
    dbert = dict(hparams("bert_base_tayp_duo_vanilla"), max_position_embeddings=128)
    gen_model_fixture("duo_bert_base_l12", r_dbert.duo_vanilla_bert_recipe, dbert, "bert", B=1, K=32, mask_seed=3407, duo=True)
    fvit = hparams("vit_base_imagenette_vanilla")   # FroyoViTConfig has the vanilla fields (no shipped froyo ViT experiment)
    gen_model_fixture("froyo_vit_base_l12", r_fvit.froyo_vit_recipe, fvit, "vit", B=1, K=32, mask_seed=3407, froyo=True)


def gen_full_depth_b2():
    """Round 6: ViT-base at full depth on TWO inputs x K = 32 = 64 masked rows — the row count from which the
    encoder runs its last layer without the K / V projection (csrc/cls_last.hip, AG_LAST_KV_SKIP default 64), so that the default
    threshold meets the reference's numbers (vit_base_l12 has 32 rows and stays on the projected form)."""
    base = hparams("vit_base_imagenette_vanilla")
    gen_model_fixture("vit_base_l12_b2", r_vvit.vanilla_vit_recipe, base, "vit", B=2, K=32, mask_seed=3407)


def gen_full_depth_aux():
    """What the full-depth tolerances are derived from: the reference run in float64 (same synthetic weights and inputs cast
    up) next to its fp32 self — |ref32 - ref64| is the reference's OWN rounding noise at 12 / 24 layers, the yardstick for
    an fp32 comparison — and the magnitude of the explainer MLP's raw output, of which phi is a small difference
    (phi = pred + ((grand - null) - sum pred) / T): a storage rounding of pred moves phi by |pred| u, not |phi| u."""
    table = {
        "vit_base_l12": (r_vvit.vanilla_vit_recipe, hparams("vit_base_imagenette_vanilla"), "vit", 32),
        "bert_base_l12": (r_vbert.vanilla_bert_recipe, dict(hparams("bert_base_tayp_vanilla"), max_position_embeddings=128), "bert", 32),
        "vit_large_l24": (r_vvit.vanilla_vit_recipe, hparams("vit_large_imagenette_vanilla"), "vit", 64),
        "duo_bert_base_l12": (r_dbert.duo_vanilla_bert_recipe, dict(hparams("bert_base_tayp_duo_vanilla"), max_position_embeddings=128), "bert", 32),
        "froyo_vit_base_l12": (r_fvit.froyo_vit_recipe, hparams("vit_base_imagenette_vanilla"), "vit", 32),
    }
    for tag, (recipe_fn, params, kind, K) in table.items():
        g = np.load(os.path.join(HERE, f"model_{tag}.npz"))
        recipe = recipe_fn()
        cfg = recipe.t_config(**params)
        P = recipe.n_players(cfg)
        m_srg, m_exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
        synth.load_synth_weights(m_srg, seed=0)
        synth.load_synth_weights(m_exp, seed=1)
        m_srg.eval(); m_exp.eval()
        if kind == "vit":
            Xs = torch.from_numpy(synth.synth_images(1, params["img_px_size"], params["img_channels"], seed=0))
        else:
            Xs = torch.from_numpy(synth.synth_token_ids(1, params["max_position_embeddings"], params["vocab_size"], seed=0))
        masks = torch.from_numpy(np.unpackbits(g["masks"], axis=-1)[:, :P].astype(np.int64))
        ones = torch.ones((1, P), dtype=torch.long)
        v_1, v_0 = torch.from_numpy(g["v_1"]), torch.from_numpy(g["v_0"])
        raw = []
        head = m_exp.explainer_mlp
        hook = head.register_forward_hook(lambda m, i, o: raw.append(o.detach()))
        with torch.no_grad():
            recipe.fw_explainer(m_exp, Xs, ones, v_1, v_0)
        hook.remove()
        pred = raw[0]
        m_srg.double(); m_exp.double()
        X64 = Xs.double() if kind == "vit" else Xs
        with torch.no_grad():
            v_s64, _ = recipe.fw_surrogate(m_srg, torch.repeat_interleave(X64, K, dim=0), masks)
            out = recipe.fw_explainer(m_exp, X64, ones, v_1.double(), v_0.double())
        phi64 = out[0]
        save(f"model_{tag}_aux.npz", v_s64=v_s64.numpy(), phi64=phi64.numpy(),
             pred_absmax=np.asarray([pred.abs().max().item()]), pred_meanabs=np.asarray([pred.abs().mean().item()]),
             noise_v_s=np.asarray([np.abs(v_s64.numpy() - g["v_s"]).max()]), noise_phi=np.asarray([np.abs(phi64.numpy() - g["phi"]).max()]))
        print(tag, "reference fp32 vs fp64: v_s", np.abs(v_s64.numpy() - g["v_s"]).max(), "phi", np.abs(phi64.numpy() - g["phi"]).max(),
              "| pred absmax", pred.abs().max().item(), "phi absmax", np.abs(g["phi"]).max())

BERT_PHI_HEAD = {"scale": {"attention.output.dense.weight": 0.3}}


def gen_bert_phi():
    """BERT-base 12 layers K=32 with an explainer whose Shapley values are NOT a cancellation artefact (round 4; see
    apply_head_tweak): max|phi| at the magnitude of max|pred|, and the reference's own autocast(bf16) deviation drops from a third
    of max|phi| (bert_base_l12) to 2 %.  Same surrogate, inputs and masks as bert_base_l12.  Also written: that autocast deviation on
    this case (the yardstick, as full_depth_bf16ref)."""
    bert = dict(hparams("bert_base_tayp_vanilla"), max_position_embeddings=128)
    tag = "bert_base_l12_phi"
    gen_model_fixture(tag, r_vbert.vanilla_bert_recipe, bert, "bert", B=1, K=32, mask_seed=3407, head=BERT_PHI_HEAD)
    g = np.load(os.path.join(HERE, f"model_{tag}.npz"))
    recipe = r_vbert.vanilla_bert_recipe()
    cfg = recipe.t_config(**bert)
    P = recipe.n_players(cfg)
    m_srg, m_exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
    synth.load_synth_weights(m_srg, seed=0)
    synth.load_synth_weights(m_exp, seed=1)
    apply_head_tweak(m_exp, BERT_PHI_HEAD)
    m_srg.eval(); m_exp.eval()
    Xs = torch.from_numpy(synth.synth_token_ids(1, bert["max_position_embeddings"], bert["vocab_size"], seed=0))
    masks = torch.from_numpy(np.unpackbits(g["masks"], axis=-1)[:, :P].astype(np.int64))
    ones = torch.ones((1, P), dtype=torch.long)
    v_1, v_0 = torch.from_numpy(g["v_1"]), torch.from_numpy(g["v_0"])
    raw = []
    hook = m_exp.explainer_mlp.register_forward_hook(lambda m, i, o: raw.append(o.detach()))
    with torch.no_grad():
        recipe.fw_explainer(m_exp, Xs, ones, v_1, v_0)
    hook.remove()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        v_s_ac, _ = recipe.fw_surrogate(m_srg, torch.repeat_interleave(Xs, 32, dim=0), masks)
        v_1_ac, _ = recipe.fw_surrogate(m_srg, Xs, ones)
        out = recipe.fw_explainer(m_exp, Xs, ones, v_1, v_0)
    phi_ac = out[0].float().numpy()
    v_s_ac, v_1_ac = v_s_ac.float().numpy(), v_1_ac.float().numpy()
    e_vs, e_phi = np.abs(v_s_ac - g["v_s"]), np.abs(phi_ac - g["phi"])
    save(f"model_{tag}_bf16ref.npz", v_s=v_s_ac, v_1=v_1_ac, phi=phi_ac,
         v_s_maxabs=np.asarray([e_vs.max()]), v_s_rms=np.asarray([np.sqrt((e_vs ** 2).mean())]),
         phi_maxabs=np.asarray([e_phi.max()]), phi_rms=np.asarray([np.sqrt((e_phi ** 2).mean())]),
         phi_absmax=np.asarray([np.abs(g["phi"]).max()]), pred_absmax=np.asarray([raw[0].abs().max().item()]))
    print(tag, "max|pred|", raw[0].abs().max().item(), "max|phi|", np.abs(g["phi"]).max(),
          "| reference under autocast(bf16): phi max", e_phi.max(), "=", e_phi.max() / np.abs(g["phi"]).max(), "of max|phi|; v_s max", e_vs.max())


def _full_depth_table():
    return {
        "vit_base_l12": (r_vvit.vanilla_vit_recipe, hparams("vit_base_imagenette_vanilla"), "vit", 32),
        "bert_base_l12": (r_vbert.vanilla_bert_recipe, dict(hparams("bert_base_tayp_vanilla"), max_position_embeddings=128), "bert", 32),
        "vit_large_l24": (r_vvit.vanilla_vit_recipe, hparams("vit_large_imagenette_vanilla"), "vit", 64),
        "duo_bert_base_l12": (r_dbert.duo_vanilla_bert_recipe, dict(hparams("bert_base_tayp_duo_vanilla"), max_position_embeddings=128), "bert", 32),
        "froyo_vit_base_l12": (r_fvit.froyo_vit_recipe, hparams("vit_base_imagenette_vanilla"), "vit", 32),
    }


def gen_full_depth_bf16ref():
    """The yardstick for the throughput mode (round 3): the REFERENCE ITSELF in its own reduced-precision mode — the same
    modules, weights, inputs and masks as model_<tag>.npz run under torch.autocast("cpu", dtype=torch.bfloat16), i.e. what a
    reference user gets from stock mixed precision (bf16 matmul operands, fp32 LayerNorm / soft-max / residual adds).  Stored:
    its v_s / v_1 / phi, so that the bf16 tests and bench.py can state the HIP bf16 mode's deviation from the fp32 reference
    NEXT TO the reference's own bf16 deviation from itself."""
    table = dict(_full_depth_table())
    if len(sys.argv) > 2 and sys.argv[2] == "ltt":
        table = _ltt_full_params()
    for tag, (recipe_fn, params, kind, K) in table.items():
        g = np.load(os.path.join(HERE, f"model_{tag}.npz"))
        recipe = recipe_fn()
        cfg = recipe.t_config(**params)
        P = recipe.n_players(cfg)
        m_srg, m_exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
        synth.load_synth_weights(m_srg, seed=0)
        synth.load_synth_weights(m_exp, seed=1)
        m_srg.eval(); m_exp.eval()
        if kind == "vit":
            Xs = torch.from_numpy(synth.synth_images(1, params["img_px_size"], params["img_channels"], seed=0))
        else:
            Xs = torch.from_numpy(synth.synth_token_ids(1, params["max_position_embeddings"], params["vocab_size"], seed=0))
        masks = torch.from_numpy(np.unpackbits(g["masks"], axis=-1)[:, :P].astype(np.int64))
        ones = torch.ones((1, P), dtype=torch.long)
        v_1, v_0 = torch.from_numpy(g["v_1"]), torch.from_numpy(g["v_0"])
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            v_s_ac, _ = recipe.fw_surrogate(m_srg, torch.repeat_interleave(Xs, K, dim=0), masks)
            v_1_ac, _ = recipe.fw_surrogate(m_srg, Xs, ones)
            out = recipe.fw_explainer(m_exp, Xs, ones, v_1, v_0)
        phi_ac = out[0].float().numpy()
        v_s_ac, v_1_ac = v_s_ac.float().numpy(), v_1_ac.float().numpy()
        e_vs, e_phi = np.abs(v_s_ac - g["v_s"]), np.abs(phi_ac - g["phi"])
        save(f"model_{tag}_bf16ref.npz", v_s=v_s_ac, v_1=v_1_ac, phi=phi_ac,
             v_s_maxabs=np.asarray([e_vs.max()]), v_s_rms=np.asarray([np.sqrt((e_vs ** 2).mean())]),
             phi_maxabs=np.asarray([e_phi.max()]), phi_rms=np.asarray([np.sqrt((e_phi ** 2).mean())]),
             phi_absmax=np.asarray([np.abs(g["phi"]).max()]))
        print(tag, "reference under autocast(bf16) vs its fp32 self: v_s max", e_vs.max(), "rms", np.sqrt((e_vs ** 2).mean()),
              "| v_1", np.abs(v_1_ac - g["v_1"]).max(), "| phi max", e_phi.max(), "=", e_phi.max() / np.abs(g["phi"]).max(), "of max|phi|")

def _ltt_full_params():
    """the shipped LTT shape (experiments/bert_base_tayp_ltt/.hparams.json: 12 layers, 96-wide ladder = 12 heads of 8, 384
    intermediate) on BERT-base at L=128 (BASELINE's sequence length) and the same ladder on ViT-base (no shipped LTT ViT
    experiment: the vanilla ViT-base backbone fields + the shipped ladder fields)."""
    lbert = dict(hparams("bert_base_tayp_ltt"), max_position_embeddings=128)
    base = hparams("vit_base_imagenette_vanilla")
    lvit = {k: v for k, v in base.items() if not k.startswith("explainer_")}
    lvit.update(explainer_s_attn_num_layers=1, explainer_s_head_hidden_size=3072, explainer_normalize=True,
                s_attn_hidden_size=96, s_attn_intermediate_size=384)
    return {"ltt_bert_base_l12": (r_lbert.ltt_bert_recipe, lbert, "bert", 32),
            "ltt_vit_base_l12": (r_lvit.ltt_vit_recipe, lvit, "vit", 32)}


def gen_ltt_full_depth():
    """LTT at the shipped / benchmarked shape: 12 backbone layers, h = 96 ladder, K = 32, one input — the
    shapes at which the ring GEMM with the ladder's map epilogue, the narrow-head MFMA attention at 6304 / 4096 rows and the
    chained LayerNorm-fold statistics across the per-layer backbone calls are selected.  Besides the model fixture
    (gen_model_fixture(ltt=True): v_0 / v_s / v_s_cls / v_1 / v_edge / phi / exp_logits / fin_logits / fin_phi) the ladder is
    pinned at intermediate depths through the reference's own knob: ``ltt_freeze_layers_until(l)`` stops the side network after
    l layers (models/ltt_vit.py:400-404, :424-426), so v_s under l = 1, 4, 8 is the head's view of the side state after l
    layers — a per-layer side-state trace through the public interface."""
    for tag, (recipe_fn, params, kind, K) in _ltt_full_params().items():
        gen_model_fixture(tag, recipe_fn, params, kind, B=1, K=K, mask_seed=3407, ltt=True)
        g = np.load(os.path.join(HERE, f"model_{tag}.npz"))
        recipe = recipe_fn()
        cfg = recipe.t_config(**params)
        P = recipe.n_players(cfg)
        m_srg = recipe.t_surrogate(cfg)
        synth.load_synth_weights(m_srg, seed=0)
        m_srg.eval()
        if kind == "vit":
            Xs = torch.from_numpy(synth.synth_images(1, params["img_px_size"], params["img_channels"], seed=0))
        else:
            Xs = torch.from_numpy(synth.synth_token_ids(1, params["max_position_embeddings"], params["vocab_size"], seed=0))
        masks = torch.from_numpy(np.unpackbits(g["masks"], axis=-1)[:, :P].astype(np.int64))
        out = {}
        enc = (m_srg.vit if kind == "vit" else m_srg.bert).encoder
        for l in (1, 4, 8):
            enc.ltt_freeze_layers_until(l)
            with torch.no_grad():
                v_l, _ = recipe.fw_surrogate(m_srg, torch.repeat_interleave(Xs, K, dim=0), masks)
            out[f"v_s_freeze{l}"] = v_l.numpy()
        enc.ltt_freeze_layers_until(params["num_hidden_layers"])
        save(f"model_{tag}_freeze.npz", **out)
        # the reference's own fp32 rounding noise at this depth (its deviation from itself in float64): the yardstick of the
        # fp32 comparison of phi, a small difference of pred-sized numbers (as gen_full_depth_aux does for the vanilla configs)
        m_exp, m_fin = recipe.t_explainer(cfg), recipe.t_final(cfg)
        synth.load_synth_weights(m_exp, seed=1)
        synth.load_synth_weights(m_fin, seed=2)
        m_srg.double().eval(); m_exp.double().eval(); m_fin.double().eval()
        X64 = Xs.double() if kind == "vit" else Xs
        ones = torch.ones((1, P), dtype=torch.long)
        with torch.no_grad():
            v_s64, _ = recipe.fw_surrogate(m_srg, torch.repeat_interleave(X64, K, dim=0), masks)
            phi64 = recipe.fw_explainer(m_exp, X64, ones, torch.from_numpy(g["v_1"]).double(), torch.from_numpy(g["v_0"]).double())[0]
            _, fin_phi64 = recipe.fw_final(m_fin, X64)
        noise = dict(noise_v_s=np.asarray([np.abs(v_s64.numpy() - g["v_s"]).max()]), noise_phi=np.asarray([np.abs(phi64.numpy() - g["phi"]).max()]),
                     noise_fin_phi=np.asarray([np.abs(fin_phi64.numpy() - g["fin_phi"]).max()]))
        save(f"model_{tag}_aux.npz", **noise)
        print(tag, "reference fp32 vs fp64:", {k_: float(v_[0]) for k_, v_ in noise.items()}, "| max|phi|", np.abs(g["phi"]).max())


def gen_perturbed_ties():
    """_get_perturbed_samples on attributions WITH ties (SURVEY §8c fixture 4).  The reference ranks with np.argsort's
    default kind, which is not stable: the order inside a tie group is whatever this host's numpy build does (AVX-512
    sorting networks here), so the recorded masks pin the reference's behaviour on THIS machine only; the fixture also
    records the stable order so that a test can show the two differ."""
    _stub_modules()
    from reference.scripts.measure_faithfulness import _get_perturbed_samples
    g = np.random.default_rng(78)
    arrs = {}
    cases = []
    for i, (P, steps) in enumerate([(196, 16), (196, 196), (127, 32)]):
        attr = (np.round(g.standard_normal(P) * 2) / 2).astype(np.float32)      # ~12 distinct values: large tie groups
        if i == 1:
            attr[:] = 0.25                                                        # one tie group of everything
        for base in (0, 1):
            stops, masks = _get_perturbed_samples(torch.from_numpy(attr), P, steps, base)
            arrs[f"c{i}_b{base}_stops"] = stops.numpy()
            arrs[f"c{i}_b{base}_masks"] = np.packbits(masks.numpy().astype(np.uint8), axis=1)
        arrs[f"c{i}_attr"] = attr
        arrs[f"c{i}_ranking_reference_host"] = np.argsort(attr)[::-1].astype(np.int64)
        arrs[f"c{i}_ranking_stable"] = np.argsort(attr, kind="stable")[::-1].astype(np.int64)
        cases.append([P, steps])
    arrs["cases"] = np.asarray(cases, dtype=np.int64)
    import platform
    with open(os.path.join(HERE, "perturbed_ties.json"), "w") as f:
        json.dump({"numpy": np.__version__, "machine": platform.machine(),
                   "note": "np.argsort default kind; tie order is host/numpy-build specific"}, f, indent=1)
    save("perturbed_ties.npz", **arrs)


def sample_idx(n):
    return np.linspace(0, n - 1, min(n, 16)).astype(np.int64)


def gen_train_step():
    """One reference _explainer_epoch_train step (SURVEY §8c fixture 5): scripts/train_explainer.py:128-207 run unmodified
    on the CPU with dropout probabilities 0, one batch, AdamW lr 1e-3; per trainable parameter the gradient the step saw
    (captured by wrapping optimizer.step) and the post-step value, as checksums + 16 samples; plus the loss and the masks."""
    _stub_modules()
    import reference.scripts.train_explainer as rte

    class Env:
        def log(self, msg):
            pass

    def one(tag, recipe_fn, params, kind, B, K, froyo=False):
        params = dict(params, attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0)
        recipe = recipe_fn()
        cfg = recipe.t_config(**params)
        P = recipe.n_players(cfg)
        m_srg, m_exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
        synth.load_synth_weights(m_srg, seed=0)
        synth.load_synth_weights(m_exp, seed=1)
        if kind == "vit":
            Xs = torch.from_numpy(synth.synth_images(B, params["img_px_size"], params["img_channels"], seed=0))
            null = torch.zeros((1, params["img_channels"], params["img_px_size"], params["img_px_size"]))
        else:
            L = params["max_position_embeddings"]
            Xs = torch.from_numpy(synth.synth_token_ids(B, L, params["vocab_size"], seed=0))
            Xs[:, -9:] = params["pad_token_id"]          # [PAD] positions: ordinary players whose embedding row gets no gradient
            null = torch.from_numpy(synth.synth_null_ids(L, params["vocab_size"]))
        m_srg.eval()
        with torch.no_grad():
            v_0, _ = recipe.fw_surrogate(m_srg, null, torch.ones((1, P), dtype=torch.long))
        lr = 1e-3
        opt = torch.optim.AdamW(m_exp.parameters(), lr=lr)
        seen = {}
        real_step = opt.step

        def step(*a, **k):
            for n, p in m_exp.named_parameters():
                seen[n] = None if p.grad is None else p.grad.detach().clone()
            return real_step(*a, **k)
        opt.step = step
        torch.manual_seed(3407)
        loss = rte._explainer_epoch_train(env=Env(), device=torch.device("cpu"), n_mask_samples=K, n_players=P,
                                          surrogate_null=v_0, d_items=[(None, None)], m_recipe=recipe, m_surrogate=m_srg,
                                          m_explainer=m_exp, optimizer=opt, epoch=1,
                                          gen_input=lambda a, b: (Xs, torch.zeros(B, dtype=torch.long)))
        torch.manual_seed(3407)
        masks = rshap.mask_shapley_new(B * K, P)
        arrs = dict(dims=np.asarray([B, K, P], dtype=np.int64), loss_mean=np.asarray([loss], dtype=np.float64),
                    masks=np.packbits(masks.numpy().astype(np.uint8), axis=1), v_0=v_0.numpy(), lr=np.asarray([lr]))
        if kind == "bert":
            arrs["ids"] = Xs.numpy()
        names = []
        for n, p in m_exp.named_parameters():
            if not p.requires_grad or seen.get(n) is None:
                continue
            gflat, pflat = seen[n].reshape(-1).double(), p.detach().reshape(-1).double()
            idx = sample_idx(gflat.numel())
            names.append(n)
            arrs["g/" + n] = np.concatenate([[gflat.sum().item(), gflat.abs().sum().item(), gflat.abs().max().item()], gflat[idx].numpy()])
            arrs["p/" + n] = np.concatenate([[pflat.sum().item(), pflat.abs().sum().item()], pflat[idx].numpy()])
        frozen = [n for n, p in m_exp.named_parameters() if not p.requires_grad]
        with open(os.path.join(HERE, f"train_step_{tag}.json"), "w") as f:
            json.dump({"kind": kind, "froyo": froyo, "params": params, "B": B, "K": K, "trained": names, "frozen": frozen,
                       "weights": {"surrogate_seed": 0, "explainer_seed": 1}, "input_seed": 0, "torch_seed": 3407}, f, indent=1)
        save(f"train_step_{tag}.npz", **arrs)

    tiny = dict(hparams("vit_tiny_imagenette_vanilla"), num_hidden_layers=2)
    one("vit_tiny_l2", r_vvit.vanilla_vit_recipe, tiny, "vit", B=2, K=4)
    one("froyo_vit_tiny_l2", r_fvit.froyo_vit_recipe, tiny, "vit", B=2, K=4, froyo=True)
    bert = dict(hparams("bert_base_tayp_vanilla"), num_hidden_layers=2, max_position_embeddings=128)
    one("bert_base_l2", r_vbert.vanilla_bert_recipe, bert, "bert", B=2, K=4)


def gen_state_keys():
    """state-dict key -> shape of every reference class on the path (names + shapes only)."""
    import reference.recipes.duo_vanilla_vit, reference.recipes.froyo_vit  # noqa: F401
    tiny = hparams("vit_tiny_imagenette_vanilla")
    bert = dict(hparams("bert_base_tayp_vanilla"), num_hidden_layers=1, max_position_embeddings=128)
    table = {"vanilla_vit": (r_vvit.vanilla_vit_recipe, tiny), "duo_vanilla_vit": (r_dvit.duo_vanilla_vit_recipe, tiny),
             "froyo_vit": (r_fvit.froyo_vit_recipe, tiny), "vanilla_bert": (r_vbert.vanilla_bert_recipe, bert),
             "duo_vanilla_bert": (r_dbert.duo_vanilla_bert_recipe, bert), "froyo_bert": (r_fbert.froyo_bert_recipe, bert)}
    out = {}
    for kind, (fn, params) in table.items():
        rec = fn()
        cfg = rec.t_config(**params)
        out[kind] = {"params": params, "roles": {
            role: {k: list(v.shape) for k, v in getattr(rec, "t_" + role)(cfg).state_dict().items()}
            for role in ("classifier", "surrogate", "explainer", "final")}}
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(out, f)
    print("wrote state_keys.json")


def gen_hparams():
    """experiment configs in the reference's on-disk format (data files of experiments/, validated here by the reference's own
    pydantic ExpConfig) + what its get_recipe makes of them: the fixtures of tests/test_host_env.py"""
    import shutil
    _stub_modules()
    from reference.scripts.resources import get_recipe
    from reference.scripts.types import ExpConfig
    os.makedirs(os.path.join(HERE, "hparams"), exist_ok=True)
    meta = {}
    for exp in ("vit_base_imagenette_vanilla", "bert_base_tayp_ltt", "bert_base_tayp_duo_vanilla"):
        src = f"/root/reference/experiments/{exp}/.hparams.json"
        shutil.copy(src, os.path.join(HERE, "hparams", f"{exp}.hparams.json"))
        with open(src) as f:
            cfg = ExpConfig.model_validate(json.load(f))
        recipe, m_cfg = get_recipe(cfg)
        meta[exp] = {"kind": cfg.net.kind, "n_players": int(recipe.n_players(m_cfg)), "config_class": type(m_cfg).__name__,
                     "train_explainer": {"epochs": cfg.train_explainer.epochs, "n_mask_samples": cfg.train_explainer.n_mask_samples,
                                         "batch_size": cfg.train_explainer.batch_size},
                     "progressive": {"surrogate": cfg.train_surrogate.EXPERIMENTAL_progressive_training,
                                     "explainer": cfg.train_explainer.EXPERIMENTAL_progressive_training}}
    with open(os.path.join(HERE, "hparams", "expected.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hparams":
        gen_hparams()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "state_keys":
        gen_state_keys()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mc_shapley":
        gen_mc_shapley()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "full_depth_b2":   # round 6
        gen_full_depth_b2()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bert_phi":   # round 4
        gen_bert_phi()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("full_depth", "full_depth_aux", "full_depth_bf16ref", "ltt_full_depth", "perturbed_ties", "train_step"):   # round-2 additions
        {"full_depth": gen_full_depth, "full_depth_aux": gen_full_depth_aux, "perturbed_ties": gen_perturbed_ties,
         "train_step": gen_train_step, "full_depth_bf16ref": gen_full_depth_bf16ref, "ltt_full_depth": gen_ltt_full_depth}[sys.argv[1]]()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ltt":   # added after the first fixture set; leaves the others untouched
        gen_ltt_models()
        gen_ltt_state_keys()
        sys.exit(0)
    seeds = gen_seeds()
    gen_masks(seeds)
    gen_shapley_fns()
    gen_perturbed()
    gen_models()
    gen_state_keys()
    gen_ltt_models()
    gen_ltt_state_keys()
    gen_mc_shapley()
    gen_full_depth()
    gen_full_depth_aux()
    gen_perturbed_ties()
    gen_train_step()
