"""Mirror of reference recipes/duo_vanilla_vit.py (dual-objective explainer on one ViT backbone)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from ..models.duo_vanilla_vit import (DuoVanillaViTClassifier, DuoVanillaViTConfig, DuoVanillaViTExplainer,
                                      DuoVanillaViTFinal, DuoVanillaViTSurrogate)
from ..utils.nnmodel import MergeStateDictRules, New, merge_state_dicts
from . import vanilla_vit as base
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training
from .vanilla_bert import FULL_MEASUREMENTS


def conv_pretrained_classifier(cfg, model):
    v = base.conv_pretrained_classifier(cfg.into(), model)
    c = DuoVanillaViTClassifier(cfg)
    merge_state_dicts(({"vit.{_}": ..., "classifier.{_}": ...}, v), into=c)
    return c


def conv_classifier_surrogate(cfg, _misc, classifier):
    s = DuoVanillaViTSurrogate(cfg)
    merge_state_dicts(({"vit.{_}": ..., "classifier.{_}": ...}, classifier), into=s)
    return s


def conv_surrogate_explainer(cfg, _misc, surrogate):
    """keeps the classification head: the duo explainer is trained on both objectives."""
    rules: MergeStateDictRules = {"vit.{_}": ..., "classifier.{_}": ...}
    for k in base.VIT_BLOCK_KEYS:
        rules[New()] = "explainer_attn.{i}." + k + ".{wb}"
    for i in (0, 1, 3, 5):
        rules[New()] = f"explainer_mlp.{i}" + ".{wb}"
    e = DuoVanillaViTExplainer(cfg)
    merge_state_dicts((rules, surrogate), into=e)
    return e


def conv_explainer_final(cfg, misc, classifier, surrogate, explainer):
    device = surrogate.vit.embeddings.cls_token.device
    nil_xs = base.gen_null(cfg.img_px_size, cfg.img_channels, device)
    nil_mask = torch.ones((1, base._n_players(cfg)), dtype=torch.long, device=device)
    surrogate.eval()
    with torch.no_grad():
        surrogate_null, _ = base.fw_surrogate(surrogate, nil_xs, nil_mask)
    final = DuoVanillaViTFinal(cfg)
    merge_state_dicts(({"{_}": "surrogate.{_}"}, surrogate), ({"{_}": "explainer.{_}"}, explainer),
                      ({"surrogate_null": ...}, {"surrogate_null": surrogate_null}), into=final)
    return final


def fw_explainer(model, xs: Tensor, mask: Tensor, surrogate_grand: Tensor, surrogate_null: Tensor):
    """-> (phi, class probabilities) (reference recipes/duo_vanilla_vit.py:196-206)."""
    xs, mask = base._fw_xs_preprocess(xs, mask)
    attr, logits = model(xs, mask, surrogate_grand, surrogate_null)
    return attr, logits


def duo_vanilla_vit_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="duo_vanilla_vit", version=RECIPE_VERSION, t_config=DuoVanillaViTConfig,
        t_classifier=DuoVanillaViTClassifier, t_surrogate=DuoVanillaViTSurrogate,
        t_explainer=DuoVanillaViTExplainer, t_final=DuoVanillaViTFinal,
        load_misc=lambda m_path, cfg: base.VanillaViTMisc(),
        conv_pretrained_classifier=conv_pretrained_classifier, conv_classifier_surrogate=conv_classifier_surrogate,
        conv_surrogate_explainer=conv_surrogate_explainer, conv_explainer_final=conv_explainer_final,
        n_players=base._n_players,
        gen_input=lambda cfg, misc, device: base.gen_input(device),
        gen_null=lambda cfg, misc, device: base.gen_null(cfg.img_px_size, cfg.img_channels, device),
        training=ModelRecipe_Training(True, True, True, exp_variant_duo=True, exp_variant_kernel_shap=False),
        fw_classifier=base.fw_classifier, fw_surrogate=base.fw_surrogate, fw_explainer=fw_explainer,
        fw_final=base.fw_final,
        measurements=ModelRecipe_Measurements(**FULL_MEASUREMENTS),
    )
