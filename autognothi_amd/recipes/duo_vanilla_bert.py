"""Mirror of reference recipes/duo_vanilla_bert.py."""
from __future__ import annotations

from torch import Tensor

from ..models.duo_vanilla_bert import (DuoVanillaBertClassifier, DuoVanillaBertConfig, DuoVanillaBertExplainer,
                                       DuoVanillaBertFinal, DuoVanillaBertSurrogate)
from ..utils.nnmodel import merge_state_dicts
from . import vanilla_bert as base
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training

_HEADS = ["bert", "bert_pooler", "classifier"]


def conv_explainer_final(cfg, misc, classifier, surrogate, explainer):
    final = DuoVanillaBertFinal(cfg)
    merge_state_dicts(({"{_}": "surrogate.{_}"}, surrogate), ({"{_}": "explainer.{_}"}, explainer),
                      ({"surrogate_null": ...}, {"surrogate_null": base.replay_null(cfg, misc, surrogate)}), into=final)
    return final


def fw_explainer(model, xs: Tensor, mask: Tensor, surrogate_grand: Tensor, surrogate_null: Tensor):
    """module returns (raw logits, phi); the recipe swaps to (phi, logits) (reference :200-213)."""
    xs, mask, tt = base._fw_xs_preprocess(xs, mask)
    logits, attr = model(xs, mask, tt, surrogate_grand, surrogate_null)
    return attr, logits


def duo_vanilla_bert_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="duo_vanilla_bert", version=RECIPE_VERSION, t_config=DuoVanillaBertConfig,
        t_classifier=DuoVanillaBertClassifier, t_surrogate=DuoVanillaBertSurrogate,
        t_explainer=DuoVanillaBertExplainer, t_final=DuoVanillaBertFinal,
        load_misc=base.load_misc,
        conv_pretrained_classifier=lambda cfg, model: base.conv_copy(
            DuoVanillaBertClassifier(cfg), base.pre_conv_bert(cfg.into(), model), _HEADS),
        conv_classifier_surrogate=lambda cfg, misc, c: base.conv_copy(DuoVanillaBertSurrogate(cfg), c, _HEADS),
        conv_surrogate_explainer=lambda cfg, misc, s: base.conv_bert_explainer(DuoVanillaBertExplainer(cfg), s, keep_heads=True),
        conv_explainer_final=conv_explainer_final,
        n_players=lambda cfg: cfg.max_position_embeddings - 1,
        gen_input=lambda cfg, misc, device: base.gen_input(cfg.max_position_embeddings, misc.tokenizer, device),
        gen_null=lambda cfg, misc, device: base.gen_null(cfg.max_position_embeddings, misc.tokenizer, device),
        training=ModelRecipe_Training(True, True, True, exp_variant_duo=True, exp_variant_kernel_shap=False),
        fw_classifier=base.fw_classifier, fw_surrogate=base.fw_surrogate, fw_explainer=fw_explainer,
        fw_final=base.fw_final,
        measurements=ModelRecipe_Measurements(**base.FULL_MEASUREMENTS),
    )
