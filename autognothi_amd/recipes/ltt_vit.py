"""Mirror of reference recipes/ltt_vit.py (LTT = ladder side-network tuning on a frozen ViT): factory, state-dict
converters and the ``fw_*`` callables, bound to ``autognothi_amd.models.ltt_vit``."""
from __future__ import annotations

import dataclasses
from typing import Any, Optional, Tuple

import torch
from torch import Tensor, nn

from ..models.ltt_vit import LttViTConfig, LttViTExplainer, LttViTFinal, LttViTSurrogate
from ..utils.nnmodel import MergeStateDictRules, New, merge_state_dicts
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training
from .vanilla_vit import VIT_BLOCK_KEYS, _fw_xs_preprocess, _n_players, conv_pretrained_classifier as _vanilla_pre_conv
from .vanilla_vit import gen_input, gen_null


@dataclasses.dataclass
class LttViTMisc:
    pass


def ltt_vit_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="ltt_vit",
        version=RECIPE_VERSION,
        t_config=LttViTConfig,
        t_classifier=LttViTSurrogate,   # sic: the classifier stage is the surrogate class (reference :36)
        t_surrogate=LttViTSurrogate,
        t_explainer=LttViTExplainer,
        t_final=LttViTFinal,
        load_misc=lambda m_path, cfg: LttViTMisc(),
        conv_pretrained_classifier=conv_pretrained_classifier,
        conv_classifier_surrogate=conv_classifier_surrogate,
        conv_surrogate_explainer=conv_surrogate_explainer,
        conv_explainer_final=conv_explainer_final,
        n_players=_n_players,
        gen_input=lambda cfg, misc, device: gen_input(device),
        gen_null=lambda cfg, misc, device: gen_null(cfg.img_px_size, cfg.img_channels, device),
        training=ModelRecipe_Training(support_classifier=True, support_surrogate=True, support_explainer=True,
                                      exp_variant_duo=False, exp_variant_kernel_shap=False),
        fw_classifier=fw_classifier,
        fw_surrogate=fw_surrogate,
        fw_explainer=fw_explainer,
        fw_final=fw_final,
        measurements=ModelRecipe_Measurements(
            verify_final_coherency=True, allow_accuracy=True, allow_faithfulness=True, allow_cls_acc=True,
            allow_performance_cls=True, allow_performance_srg_exp=True, allow_performance_fin=True,
            allow_train_resources=True, allow_dual_task_similarity=False, allow_branches_cka=True),
    )


# ------------------------------------------------------------------ converters (reference :84-224)
def _side_rules(src_branch: int, dst_branch: Optional[int], action: Any) -> MergeStateDictRules:
    """Rules over the ladder of one branch.  action: ``...`` keep, ``None`` drop, "new" fresh, "move" -> dst_branch."""
    rules: MergeStateDictRules = {}
    keys = [f"vit.encoder.s_attn_maps.{{b}}_{{i}}.{{wb}}"] + [f"vit.encoder.s_attn_layers.{{b}}_{{i}}.{k}.{{wb}}" for k in VIT_BLOCK_KEYS]
    keys.append("vit.s_attn_layernorm.{b}.{wb}")
    for k in keys:
        src = k.replace("{b}", str(src_branch))
        if action == "new":
            rules[New()] = src
        elif action == "move":
            rules[src] = k.replace("{b}", str(dst_branch))
        else:
            rules[src] = action
    return rules


def conv_pretrained_classifier(cfg: LttViTConfig, model: Any) -> LttViTSurrogate:
    """pretrained ViT -> backbone of the LTT surrogate; the whole ladder and its head are fresh (reference :84-105)."""
    v_classifier = _vanilla_pre_conv(cfg.into(), model)
    rules: MergeStateDictRules = {"vit.embeddings.{_}": ..., "vit.encoder.layers.{_}": ..., "vit.layernorm.{wb}": ...,
                                  "classifier.{_}": ...}
    rules.update(_side_rules(0, None, "new"))
    rules[New()] = "s_attn_classifier.{wb}"
    classifier = LttViTSurrogate(cfg)
    merge_state_dicts((rules, v_classifier), into=classifier)
    return classifier


def conv_classifier_surrogate(cfg: LttViTConfig, _misc, classifier: LttViTSurrogate) -> LttViTSurrogate:
    surrogate = LttViTSurrogate(cfg)
    merge_state_dicts(({"vit.{_}": ..., "classifier.{_}": ..., "s_attn_classifier.{_}": ...}, classifier), into=surrogate)
    return surrogate


def conv_surrogate_explainer(cfg: LttViTConfig, _misc, surrogate: LttViTSurrogate) -> LttViTExplainer:
    rules: MergeStateDictRules = {"vit.{_}": ..., "classifier.{_}": ..., "s_attn_classifier.{wb}": None}
    for k in VIT_BLOCK_KEYS:   # the reference lists only the MLP as new (:130-133); a config with explainer layers needs these too
        rules[New()] = "s_explainer_attn.{i}." + k + ".{wb}"
    for i in (0, 1, 3, 5):
        rules[New()] = f"s_explainer_mlp.{i}" + ".{wb}"
    explainer = LttViTExplainer(cfg)
    merge_state_dicts((rules, surrogate), into=explainer)
    return explainer


def conv_explainer_final(cfg: LttViTConfig, misc, classifier: LttViTSurrogate, surrogate: LttViTSurrogate,
                         explainer: LttViTExplainer) -> LttViTFinal:
    """backbone from the classifier, ladder 0 + its head from the surrogate, the explainer's ladder moved to branch 1
    (reference :139-224); ``surrogate_null`` replayed on the null input."""
    device = classifier.vit.embeddings.cls_token.device
    nil_xs = gen_null(cfg.img_px_size, cfg.img_channels, device)
    nil_mask = torch.ones((1, _n_players(cfg)), dtype=torch.long, device=device)
    surrogate.eval()
    with torch.no_grad():
        surrogate_null, _ = fw_surrogate(surrogate, nil_xs, nil_mask)
    backbone: MergeStateDictRules = {"vit.embeddings.{_}": ..., "vit.encoder.layers.{_}": ..., "vit.layernorm.{wb}": ...,
                                     "classifier.{wb}": ...}
    drop_backbone: MergeStateDictRules = {"vit.embeddings.{_}": None, "vit.encoder.layers.{_}": None, "vit.layernorm.{wb}": None,
                                          "classifier.{_}": None}
    rules_cls = dict(backbone)
    rules_cls.update(_side_rules(0, None, None))
    rules_cls["s_attn_classifier.{wb}"] = None
    rules_srg = dict(drop_backbone)
    rules_srg.update(_side_rules(0, None, ...))
    rules_srg["s_attn_classifier.{wb}"] = ...
    rules_exp = dict(drop_backbone)
    rules_exp.update(_side_rules(0, 1, "move"))
    rules_exp["s_explainer_attn.{_}"] = ...
    rules_exp["s_explainer_mlp.{_}"] = ...
    final = LttViTFinal(cfg)
    merge_state_dicts((rules_cls, classifier), (rules_srg, surrogate), (rules_exp, explainer),
                      ({"surrogate_null": ...}, {"surrogate_null": surrogate_null}), into=final)
    return final


# ------------------------------------------------------------------ forwards (reference :227-268)
def fw_classifier(model: LttViTSurrogate, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    xs, mask = _fw_xs_preprocess(xs, mask)
    side_logits, logits = model(xs, mask)
    return side_logits, logits


def fw_surrogate(model: LttViTSurrogate, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    xs, mask = _fw_xs_preprocess(xs, mask)
    side_logits, logits = model(xs, mask)
    return side_logits, logits


def fw_explainer(model: LttViTExplainer, xs: Tensor, mask: Tensor, surrogate_grand: Tensor,
                 surrogate_null: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    xs, mask = _fw_xs_preprocess(xs, mask)
    side_attr, logits = model(xs, mask, surrogate_grand, surrogate_null)
    return side_attr, logits


def fw_final(model: LttViTFinal, xs: Tensor) -> Tuple[Tensor, Tensor]:
    mask = torch.ones((xs.shape[0], 1 + _n_players(model.config)), dtype=torch.long, device=xs.device)
    logits, attr = model(xs, mask)
    return logits, attr
