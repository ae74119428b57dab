"""Mirror of reference recipes/froyo_bert.py."""
from __future__ import annotations

from ..models.froyo_bert import (FroyoBertClassifier, FroyoBertConfig, FroyoBertExplainer, FroyoBertFinal,
                                 FroyoBertSurrogate)
from ..utils.nnmodel import merge_state_dicts
from . import vanilla_bert as base
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training

_HEADS = ["bert", "bert_pooler", "classifier"]


def conv_explainer_final(cfg, misc, classifier, surrogate, explainer):
    final = FroyoBertFinal(cfg)
    merge_state_dicts(
        ({"bert.{_}": ..., "bert_pooler.{_}": ..., "classifier.{_}": ...}, classifier),
        ({"bert.{_}": None, "bert_pooler.{_}": "srg_bert_pooler.{_}", "classifier.{_}": "srg_classifier.{_}"}, surrogate),
        ({"bert.{_}": None, "explainer_attn.{_}": ..., "explainer_mlp.{_}": ...}, explainer),
        ({"surrogate_null": ...}, {"surrogate_null": base.replay_null(cfg, misc, surrogate)}), into=final)
    return final


def froyo_bert_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="froyo_bert", version=RECIPE_VERSION, t_config=FroyoBertConfig,
        t_classifier=FroyoBertClassifier, t_surrogate=FroyoBertSurrogate, t_explainer=FroyoBertExplainer,
        t_final=FroyoBertFinal,
        load_misc=base.load_misc,
        conv_pretrained_classifier=lambda cfg, model: base.conv_copy(
            FroyoBertClassifier(cfg), base.pre_conv_bert(cfg.into(), model), _HEADS),
        conv_classifier_surrogate=lambda cfg, misc, c: base.conv_copy(FroyoBertSurrogate(cfg), c, _HEADS),
        conv_surrogate_explainer=lambda cfg, misc, s: base.conv_bert_explainer(FroyoBertExplainer(cfg), s, keep_heads=False),
        conv_explainer_final=conv_explainer_final,
        n_players=lambda cfg: cfg.max_position_embeddings - 1,
        gen_input=lambda cfg, misc, device: base.gen_input(cfg.max_position_embeddings, misc.tokenizer, device),
        gen_null=lambda cfg, misc, device: base.gen_null(cfg.max_position_embeddings, misc.tokenizer, device),
        training=ModelRecipe_Training(True, True, True, exp_variant_duo=False, exp_variant_kernel_shap=False),
        fw_classifier=base.fw_classifier, fw_surrogate=base.fw_surrogate, fw_explainer=base.fw_explainer,
        fw_final=base.fw_final,
        measurements=ModelRecipe_Measurements(**base.FULL_MEASUREMENTS),
    )
