"""Mirror of reference recipes/froyo_vit.py (frozen backbone; Final shares one backbone pass)."""
from __future__ import annotations

import torch

from ..models.froyo_vit import (FroyoViTClassifier, FroyoViTConfig, FroyoViTExplainer, FroyoViTFinal, FroyoViTSurrogate)
from ..utils.nnmodel import MergeStateDictRules, New, merge_state_dicts
from . import vanilla_vit as base
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training
from .vanilla_bert import FULL_MEASUREMENTS


def conv_pretrained_classifier(cfg, model):
    v = base.conv_pretrained_classifier(cfg.into(), model)
    c = FroyoViTClassifier(cfg)
    merge_state_dicts(({"vit.{_}": ..., "classifier.{_}": ...}, v), into=c)
    return c


def conv_classifier_surrogate(cfg, _misc, classifier):
    s = FroyoViTSurrogate(cfg)
    merge_state_dicts(({"vit.{_}": ..., "classifier.{_}": ...}, classifier), into=s)
    return s


def conv_surrogate_explainer(cfg, _misc, surrogate):
    rules: MergeStateDictRules = {"vit.{_}": ..., "classifier.{_}": None}
    for k in base.VIT_BLOCK_KEYS:
        rules[New()] = "explainer_attn.{i}." + k + ".{wb}"
    for i in (0, 1, 3, 5):
        rules[New()] = f"explainer_mlp.{i}" + ".{wb}"
    e = FroyoViTExplainer(cfg)
    merge_state_dicts((rules, surrogate), into=e)
    return e


def conv_explainer_final(cfg, misc, classifier, surrogate, explainer):
    """classifier keeps the backbone; the surrogate contributes only its head (as srg_classifier) and the
    explainer only its heads (reference :160-196)."""
    device = classifier.vit.embeddings.cls_token.device
    nil_xs = base.gen_null(cfg.img_px_size, cfg.img_channels, device)
    nil_mask = torch.ones((1, base._n_players(cfg)), dtype=torch.long, device=device)
    surrogate.eval()
    with torch.no_grad():
        surrogate_null, _ = base.fw_surrogate(surrogate, nil_xs, nil_mask)
    final = FroyoViTFinal(cfg)
    merge_state_dicts(({"vit.{_}": ..., "classifier.{_}": ...}, classifier),
                      ({"vit.{_}": None, "classifier.{_}": "srg_classifier.{_}"}, surrogate),
                      ({"vit.{_}": None, "explainer_attn.{_}": ..., "explainer_mlp.{_}": ...}, explainer),
                      ({"surrogate_null": ...}, {"surrogate_null": surrogate_null}), into=final)
    return final


def froyo_vit_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="froyo_vit", version=RECIPE_VERSION, t_config=FroyoViTConfig,
        t_classifier=FroyoViTClassifier, t_surrogate=FroyoViTSurrogate, t_explainer=FroyoViTExplainer,
        t_final=FroyoViTFinal,
        load_misc=lambda m_path, cfg: base.VanillaViTMisc(),
        conv_pretrained_classifier=conv_pretrained_classifier, conv_classifier_surrogate=conv_classifier_surrogate,
        conv_surrogate_explainer=conv_surrogate_explainer, conv_explainer_final=conv_explainer_final,
        n_players=base._n_players,
        gen_input=lambda cfg, misc, device: base.gen_input(device),
        gen_null=lambda cfg, misc, device: base.gen_null(cfg.img_px_size, cfg.img_channels, device),
        training=ModelRecipe_Training(True, True, True, exp_variant_duo=False, exp_variant_kernel_shap=False),
        fw_classifier=base.fw_classifier, fw_surrogate=base.fw_surrogate, fw_explainer=base.fw_explainer,
        fw_final=base.fw_final,
        measurements=ModelRecipe_Measurements(**FULL_MEASUREMENTS),
    )
