"""Mirror of reference recipes/vanilla_vit.py: the factory, the state-dict converters and the four
``fw_*`` callables, bound to the HIP-backed modules of ``autognothi_amd.models.vanilla_vit``."""
from __future__ import annotations

import dataclasses
import pathlib
from typing import Any, Callable, List, Optional, Tuple

import torch
from torch import Tensor, nn

from ..models.vanilla_vit import (VanillaViTClassifier, VanillaViTConfig, VanillaViTExplainer, VanillaViTFinal,
                                  VanillaViTSurrogate)
from ..utils.nnmodel import MergeStateDictRules, New, merge_state_dicts
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training


@dataclasses.dataclass
class VanillaViTMisc:
    pass


def _n_players(cfg) -> int:
    return (cfg.img_px_size // cfg.img_patch_size) ** 2


def vanilla_vit_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="vanilla_bert",  # sic: the reference's id string (recipes/vanilla_vit.py:37), only used in logs
        version=RECIPE_VERSION,
        t_config=VanillaViTConfig,
        t_classifier=VanillaViTClassifier,
        t_surrogate=VanillaViTSurrogate,
        t_explainer=VanillaViTExplainer,
        t_final=VanillaViTFinal,
        load_misc=lambda m_path, cfg: VanillaViTMisc(),
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


# ------------------------------------------------------------------ converters (reference :86-196)
VIT_BLOCK_KEYS = ["attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                  "intermediate.dense", "output.dense", "layernorm_before", "layernorm_after"]


def conv_pretrained_classifier(cfg: VanillaViTConfig, model: Any) -> VanillaViTClassifier:
    """Own-format classifier -> classifier, or a HF ``ViTForImageClassification``-style state dict
    (``vit.encoder.layer.{i}.attention.attention.query`` naming) with a fresh head."""
    sd = model.state_dict() if isinstance(model, nn.Module) else model
    if any(k.startswith("vit.encoder.layer.") for k in sd):
        rules: MergeStateDictRules = {
            "vit.embeddings.cls_token": ..., "vit.embeddings.position_embeddings": ...,
            "vit.embeddings.patch_embeddings.projection.{wb}": ...,
            "vit.encoder.layer.{i}.attention.attention.query.{wb}": "vit.encoder.layers.{i}.attention.self.query.{wb}",
            "vit.encoder.layer.{i}.attention.attention.key.{wb}": "vit.encoder.layers.{i}.attention.self.key.{wb}",
            "vit.encoder.layer.{i}.attention.attention.value.{wb}": "vit.encoder.layers.{i}.attention.self.value.{wb}",
            "vit.encoder.layer.{i}.attention.output.dense.{wb}": "vit.encoder.layers.{i}.attention.output.dense.{wb}",
            "vit.encoder.layer.{i}.intermediate.dense.{wb}": "vit.encoder.layers.{i}.intermediate.dense.{wb}",
            "vit.encoder.layer.{i}.output.dense.{wb}": "vit.encoder.layers.{i}.output.dense.{wb}",
            "vit.encoder.layer.{i}.layernorm_before.{wb}": "vit.encoder.layers.{i}.layernorm_before.{wb}",
            "vit.encoder.layer.{i}.layernorm_after.{wb}": "vit.encoder.layers.{i}.layernorm_after.{wb}",
            "vit.layernorm.{wb}": ..., "classifier.{wb}": None, New(): "classifier.{wb}",
        }
    else:
        rules = {"{_}": ...}
    classifier = VanillaViTClassifier(cfg)
    merge_state_dicts((rules, sd), into=classifier)
    return classifier


def conv_classifier_surrogate(cfg: VanillaViTConfig, _misc, classifier: VanillaViTClassifier) -> VanillaViTSurrogate:
    surrogate = VanillaViTSurrogate(cfg)
    merge_state_dicts(({"vit.{_}": ..., "classifier.{_}": ...}, classifier), into=surrogate)
    return surrogate


def conv_surrogate_explainer(cfg: VanillaViTConfig, _misc, surrogate: VanillaViTSurrogate) -> VanillaViTExplainer:
    rules: MergeStateDictRules = {"vit.{_}": ..., "classifier.{_}": None}
    for k in VIT_BLOCK_KEYS:
        rules[New()] = "explainer_attn.{i}." + k + ".{wb}"
    for i in (0, 1, 3, 5):
        rules[New()] = f"explainer_mlp.{i}" + ".{wb}"
    explainer = VanillaViTExplainer(cfg)
    merge_state_dicts((rules, surrogate), into=explainer)
    return explainer


def conv_explainer_final(cfg: VanillaViTConfig, misc, classifier, surrogate, explainer) -> VanillaViTFinal:
    """Replays the surrogate on the null input to freeze ``surrogate_null`` (reference :160-196)."""
    device = classifier.vit.embeddings.cls_token.device
    nil_xs = gen_null(cfg.img_px_size, cfg.img_channels, device)
    nil_mask = torch.ones((1, _n_players(cfg)), dtype=torch.long, device=device)
    surrogate.eval()
    with torch.no_grad():
        surrogate_null, _ = fw_surrogate(surrogate, nil_xs, nil_mask)
    final = VanillaViTFinal(cfg)
    merge_state_dicts(({"{_}": "classifier.{_}"}, classifier), ({"{_}": "surrogate.{_}"}, surrogate),
                      ({"{_}": "explainer.{_}"}, explainer), ({"surrogate_null": ...}, {"surrogate_null": surrogate_null}),
                      into=final)
    return final


# ------------------------------------------------------------------ inputs
def gen_input(device: torch.device) -> Callable[[Any, Any], Tuple[Tensor, Tensor]]:
    def collate(raw_xs: List[Tensor], raw_ys: List[int]):
        return torch.stack(raw_xs, dim=0).to(device), torch.tensor(raw_ys).to(device)
    return collate


def gen_null(img_px_size: int, img_channels: int, device: torch.device) -> Tensor:
    """zero image (reference :213-216)."""
    return torch.zeros((1, img_channels, img_px_size, img_px_size), device=device)


# ------------------------------------------------------------------ forwards (reference :219-261)
def _fw_xs_preprocess(xs: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """Prepend the always-on CLS column.  ``mask`` [R,P] int64 (or pre-packed key bits, passed through)."""
    if mask.dtype == torch.int32:
        return xs, mask
    mask_cls = torch.ones((mask.shape[0], 1), dtype=mask.dtype, device=mask.device)
    return xs, torch.cat([mask_cls, mask], dim=1)


def fw_classifier(model: VanillaViTClassifier, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    xs, mask = _fw_xs_preprocess(xs, mask)
    logits = model(xs, mask)
    return logits, logits


def fw_surrogate(model: VanillaViTSurrogate, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    xs, mask = _fw_xs_preprocess(xs, mask)
    return model(xs, mask), None


def fw_explainer(model: VanillaViTExplainer, xs: Tensor, mask: Tensor, surrogate_grand: Tensor,
                 surrogate_null: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    xs, mask = _fw_xs_preprocess(xs, mask)
    return model(xs, mask, surrogate_grand, surrogate_null), None


def fw_final(model: VanillaViTFinal, xs: Tensor) -> Tuple[Tensor, Tensor]:
    n_players = _n_players(model.config)
    mask = torch.ones((xs.shape[0], 1 + n_players), dtype=torch.long, device=xs.device)
    return model(xs, mask)
