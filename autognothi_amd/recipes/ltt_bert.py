"""Mirror of reference recipes/ltt_bert.py (ladder side-network tuning on a frozen BERT), bound to
``autognothi_amd.models.ltt_bert``."""
from __future__ import annotations

import dataclasses
from typing import Any, Optional, Tuple

import torch
from torch import Tensor

from ..models.ltt_bert import LttBertConfig, LttBertExplainer, LttBertFinal, LttBertSurrogate
from ..utils.nnmodel import MergeStateDictRules, New, merge_state_dicts
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training
from .vanilla_bert import BERT_BLOCK_KEYS, FULL_MEASUREMENTS, _fw_xs_preprocess, gen_input, gen_null, load_misc as _load_tok, pre_conv_bert


@dataclasses.dataclass
class LttBertMisc:
    tokenizer: Any


def ltt_bert_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="ltt_bert", version=RECIPE_VERSION, t_config=LttBertConfig,
        t_classifier=LttBertSurrogate,   # sic: the classifier stage is the surrogate class (reference :40)
        t_surrogate=LttBertSurrogate, t_explainer=LttBertExplainer, t_final=LttBertFinal,
        load_misc=lambda m_path, cfg: LttBertMisc(tokenizer=_load_tok(m_path, cfg).tokenizer),
        conv_pretrained_classifier=conv_pretrained_classifier,
        conv_classifier_surrogate=conv_classifier_surrogate,
        conv_surrogate_explainer=conv_surrogate_explainer,
        conv_explainer_final=conv_explainer_final,
        n_players=lambda cfg: cfg.max_position_embeddings - 1,
        gen_input=lambda cfg, misc, device: gen_input(cfg.max_position_embeddings, misc.tokenizer, device),
        gen_null=lambda cfg, misc, device: gen_null(cfg.max_position_embeddings, misc.tokenizer, device),
        training=ModelRecipe_Training(True, True, True, exp_variant_duo=False, exp_variant_kernel_shap=False),
        fw_classifier=fw_classifier, fw_surrogate=fw_surrogate, fw_explainer=fw_explainer, fw_final=fw_final,
        measurements=ModelRecipe_Measurements(**FULL_MEASUREMENTS),
    )


# ------------------------------------------------------------------ converters (reference :92-254)
def _side_rules(src_branch: int, dst_branch: Optional[int], action: Any) -> MergeStateDictRules:
    """Rules over the ladder of one branch.  action: ``...`` keep, ``None`` drop, "new" fresh, "move" -> dst_branch."""
    rules: MergeStateDictRules = {}
    keys = ["bert.encoder.s_attn_maps.{b}_{i}.{wb}"] + [f"bert.encoder.s_attn_layers.{{b}}_{{i}}.{k}.{{wb}}" for k in BERT_BLOCK_KEYS]
    for k in keys:
        src = k.replace("{b}", str(src_branch))
        if action == "new":
            rules[New()] = src
        elif action == "move":
            rules[src] = k.replace("{b}", str(dst_branch))
        else:
            rules[src] = action
    return rules


def conv_pretrained_classifier(cfg: LttBertConfig, model: Any) -> LttBertSurrogate:
    v_classifier = pre_conv_bert(cfg.into(), model)
    rules: MergeStateDictRules = {"bert.embeddings.{_}": ..., "bert.encoder.layers.{_}": ..., "bert_pooler.dense.{wb}": ...,
                                  "classifier.{wb}": ...}
    rules.update(_side_rules(0, None, "new"))
    rules[New()] = "bert_s_attn_pooler.dense.{wb}"
    rules[New()] = "s_attn_classifier.{wb}"
    classifier = LttBertSurrogate(cfg)
    merge_state_dicts((rules, v_classifier), into=classifier)
    return classifier


def conv_classifier_surrogate(cfg: LttBertConfig, _misc, classifier: LttBertSurrogate) -> LttBertSurrogate:
    rules: MergeStateDictRules = {"bert.{_}": ..., "bert_pooler.{_}": ..., "classifier.{_}": ..., "bert_s_attn_pooler.{_}": ...,
                                  "s_attn_classifier.{_}": ...}
    surrogate = LttBertSurrogate(cfg)
    merge_state_dicts((rules, classifier), into=surrogate)
    return surrogate


def conv_surrogate_explainer(cfg: LttBertConfig, _misc, surrogate: LttBertSurrogate) -> LttBertExplainer:
    rules: MergeStateDictRules = {"bert.{_}": ..., "bert_pooler.{_}": ..., "bert_s_attn_pooler.{_}": None, "classifier.{_}": ...,
                                  "s_attn_classifier.{wb}": None}
    for k in BERT_BLOCK_KEYS:
        rules[New()] = "s_attn_attention_layers.{i}." + k + ".{wb}"
    for i in (0, 2, 4):
        rules[New()] = f"s_attn_explainer.{i}" + ".{wb}"
    explainer = LttBertExplainer(cfg)
    merge_state_dicts((rules, surrogate), into=explainer)
    return explainer


def conv_explainer_final(cfg: LttBertConfig, misc: LttBertMisc, classifier: LttBertSurrogate, surrogate: LttBertSurrogate,
                         explainer: LttBertExplainer) -> LttBertFinal:
    device = classifier.bert.embeddings.word_embeddings.weight.device
    nil_xs = gen_null(cfg.max_position_embeddings, misc.tokenizer, device)
    nil_mask = torch.ones((1, cfg.max_position_embeddings - 1), dtype=torch.long, device=device)
    surrogate.eval()
    with torch.no_grad():
        surrogate_null, _ = fw_surrogate(surrogate, nil_xs, nil_mask)
    backbone: MergeStateDictRules = {"bert.embeddings.{_}": ..., "bert.encoder.layers.{_}": ..., "bert_pooler.dense.{wb}": ...,
                                     "classifier.{wb}": ...}
    drop_backbone: MergeStateDictRules = {"bert.embeddings.{_}": None, "bert.encoder.layers.{_}": None, "bert_pooler.{_}": None,
                                          "classifier.{_}": None}
    rules_cls = dict(backbone)
    rules_cls.update(_side_rules(0, None, None))
    rules_cls["bert_s_attn_pooler.dense.{wb}"] = None
    rules_cls["s_attn_classifier.{wb}"] = None
    rules_srg = dict(drop_backbone)
    rules_srg.update(_side_rules(0, None, ...))
    rules_srg["bert_s_attn_pooler.dense.{wb}"] = ...
    rules_srg["s_attn_classifier.{wb}"] = ...
    rules_exp = dict(drop_backbone)
    rules_exp.update(_side_rules(0, 1, "move"))
    rules_exp["s_attn_attention_layers.{_}"] = ...
    rules_exp["s_attn_explainer.{_}"] = ...
    final = LttBertFinal(cfg)
    merge_state_dicts((rules_cls, classifier), (rules_srg, surrogate), (rules_exp, explainer),
                      ({"surrogate_null": ...}, {"surrogate_null": surrogate_null}), into=final)
    return final


# ------------------------------------------------------------------ forwards (reference :257-297)
def fw_classifier(model: LttBertSurrogate, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    xs, mask, tt = _fw_xs_preprocess(xs, mask)
    side_logits, logits = model(xs, mask, tt)
    return side_logits, logits


def fw_surrogate(model: LttBertSurrogate, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    xs, mask, tt = _fw_xs_preprocess(xs, mask)
    side_logits, logits = model(xs, mask, tt)
    return side_logits, logits


def fw_explainer(model: LttBertExplainer, xs: Tensor, mask: Tensor, surrogate_grand: Tensor,
                 surrogate_null: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    xs, mask, tt = _fw_xs_preprocess(xs, mask)
    side_attr, logits = model(xs, mask, tt, surrogate_grand, surrogate_null)
    return side_attr, logits


def fw_final(model: LttBertFinal, xs: Tensor) -> Tuple[Tensor, Tensor]:
    mask = torch.ones_like(xs)
    logits, attr = model(xs, mask, None)
    return logits, attr
