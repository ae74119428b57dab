"""Mirror of reference recipes/vanilla_bert.py bound to the HIP-backed BERT modules."""
from __future__ import annotations

import dataclasses
import pathlib
from typing import Any, Callable, List, Optional, Tuple

import torch
from torch import Tensor, nn

from ..models.vanilla_bert import (VanillaBertClassifier, VanillaBertConfig, VanillaBertExplainer, VanillaBertFinal,
                                   VanillaBertSurrogate)
from ..utils.nnmodel import MergeStateDictRules, New, merge_state_dicts
from .types import RECIPE_VERSION, ModelRecipe, ModelRecipe_Measurements, ModelRecipe_Training


@dataclasses.dataclass
class VanillaBertMisc:
    tokenizer: Any


BERT_BLOCK_KEYS = ["attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                   "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"]

FULL_MEASUREMENTS = dict(verify_final_coherency=True, allow_accuracy=True, allow_faithfulness=True, allow_cls_acc=True,
                         allow_performance_cls=True, allow_performance_srg_exp=True, allow_performance_fin=True,
                         allow_train_resources=True, allow_dual_task_similarity=False, allow_branches_cka=True)


def load_misc(m_path: pathlib.Path, cfg) -> VanillaBertMisc:
    from transformers import AutoTokenizer  # host-side tokenisation, as the reference (:90-94)
    return VanillaBertMisc(tokenizer=AutoTokenizer.from_pretrained(pathlib.Path(m_path) / "tokenizer"))


def vanilla_bert_recipe() -> ModelRecipe:
    return ModelRecipe(
        id="vanilla_bert", version=RECIPE_VERSION, t_config=VanillaBertConfig,
        t_classifier=VanillaBertClassifier, t_surrogate=VanillaBertSurrogate,
        t_explainer=VanillaBertExplainer, t_final=VanillaBertFinal,
        load_misc=load_misc,
        conv_pretrained_classifier=lambda cfg, model: pre_conv_bert(cfg, model, VanillaBertClassifier),
        conv_classifier_surrogate=lambda cfg, misc, c: conv_copy(VanillaBertSurrogate(cfg), c, ["bert", "bert_pooler", "classifier"]),
        conv_surrogate_explainer=lambda cfg, misc, s: conv_bert_explainer(VanillaBertExplainer(cfg), s, keep_heads=False),
        conv_explainer_final=conv_explainer_final,
        n_players=lambda cfg: cfg.max_position_embeddings - 1,
        gen_input=lambda cfg, misc, device: gen_input(cfg.max_position_embeddings, misc.tokenizer, device),
        gen_null=lambda cfg, misc, device: gen_null(cfg.max_position_embeddings, misc.tokenizer, device),
        training=ModelRecipe_Training(True, True, True, exp_variant_duo=False, exp_variant_kernel_shap=False),
        fw_classifier=fw_classifier, fw_surrogate=fw_surrogate, fw_explainer=fw_explainer, fw_final=fw_final,
        measurements=ModelRecipe_Measurements(**FULL_MEASUREMENTS),
    )


# ------------------------------------------------------------------ converters (reference :97-226)
def pre_conv_bert(cfg, model: Any, t_classifier=VanillaBertClassifier):
    sd = model.state_dict() if isinstance(model, nn.Module) else model
    blocks = {f"{{p}}encoder.layer.{{i}}.{k}.{{wb}}": f"bert.encoder.layers.{{i}}.{k}.{{wb}}" for k in BERT_BLOCK_KEYS}
    if any(k.startswith("bert.encoder.layer.") for k in sd):      # BertForSequenceClassification
        rules: MergeStateDictRules = {"bert.embeddings.{_}": ..., "bert.pooler.dense.{wb}": "bert_pooler.dense.{wb}",
                                      "classifier.{wb}": ...}
        rules.update({k.replace("{p}", "bert."): v for k, v in blocks.items()})
    elif any(k.startswith("encoder.layer.") for k in sd):         # bare BertModel: fresh head
        rules = {"embeddings.{_}": "bert.embeddings.{_}", "pooler.dense.{wb}": "bert_pooler.dense.{wb}",
                 New(): "classifier.{wb}"}
        rules.update({k.replace("{p}", ""): v for k, v in blocks.items()})
    else:
        rules = {"{_}": ...}
    sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}  # HF keeps this buffer persistent
    classifier = t_classifier(cfg)
    merge_state_dicts((rules, sd), into=classifier)
    return classifier


def conv_copy(into: nn.Module, src: nn.Module, prefixes: List[str]) -> nn.Module:
    merge_state_dicts(({p + ".{_}": ... for p in prefixes}, src), into=into)
    return into


def conv_bert_explainer(explainer: nn.Module, surrogate: nn.Module, keep_heads: bool) -> nn.Module:
    """vanilla: drop the surrogate's pooler/head (:172-174); duo: keep them (recipes/duo_vanilla_bert.py:131-134)."""
    rules: MergeStateDictRules = {"bert.{_}": ..., "bert_pooler.{_}": ... if keep_heads else None,
                                  "classifier.{_}": ... if keep_heads else None}
    for k in BERT_BLOCK_KEYS:
        rules[New()] = "explainer_attn.{i}." + k + ".{wb}"
    for i in (0, 2, 4):
        rules[New()] = f"explainer_mlp.{i}" + ".{wb}"
    merge_state_dicts((rules, surrogate), into=explainer)
    return explainer


def replay_null(cfg, misc, surrogate) -> Tensor:
    device = surrogate.bert.embeddings.word_embeddings.weight.device
    nil_xs = gen_null(cfg.max_position_embeddings, misc.tokenizer, device)
    nil_mask = torch.ones((1, cfg.max_position_embeddings - 1), dtype=torch.long, device=device)
    surrogate.eval()
    with torch.no_grad():
        surrogate_null, _ = fw_surrogate(surrogate, nil_xs, nil_mask)
    return surrogate_null


def conv_explainer_final(cfg, misc, classifier, surrogate, explainer) -> VanillaBertFinal:
    final = VanillaBertFinal(cfg)
    merge_state_dicts(({"{_}": "classifier.{_}"}, classifier), ({"{_}": "surrogate.{_}"}, surrogate),
                      ({"{_}": "explainer.{_}"}, explainer),
                      ({"surrogate_null": ...}, {"surrogate_null": replay_null(cfg, misc, surrogate)}), into=final)
    return final


# ------------------------------------------------------------------ inputs (reference :226-278)
def gen_input(max_position_embeddings: int, tokenizer: Any, device: torch.device) -> Callable[[Any, Any], Tuple[Tensor, Tensor]]:
    """[PAD] positions are ordinary players: the tokenizer's own attention mask is discarded (:250-254)."""
    def collate(raw_xs: List[str], raw_ys: List[int]):
        rows = []
        for raw_x in raw_xs:
            enc = tokenizer(raw_x, return_tensors="pt", padding="max_length", max_length=max_position_embeddings)
            rows.append(enc["input_ids"][:, :max_position_embeddings])
        return torch.cat(rows, dim=0).to(device), torch.tensor(raw_ys).to(device)
    return collate


def gen_null(max_position_embeddings: int, tokenizer: Any, device: torch.device) -> Tensor:
    enc = tokenizer("", return_tensors="pt", padding="max_length", max_length=max_position_embeddings)
    return enc["input_ids"].to(device)


# ------------------------------------------------------------------ forwards (reference :281-329)
def _fw_xs_preprocess(xs: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """CLS column prepended; token_type_ids are all zero (:289) — passed as None, which the modules
    treat as zeros without allocating them; inputs themselves are never masked (:282-288)."""
    if mask.dtype == torch.int32:
        return xs, mask, None
    mask_cls = torch.ones((mask.shape[0], 1), dtype=mask.dtype, device=mask.device)
    return xs, torch.cat([mask_cls, mask], dim=1), None


def fw_classifier(model, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    xs, mask, tt = _fw_xs_preprocess(xs, mask)
    logits = model(xs, mask, tt)
    return logits, logits


def fw_surrogate(model, xs: Tensor, mask: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    xs, mask, tt = _fw_xs_preprocess(xs, mask)
    return model(xs, mask, tt), None


def fw_explainer(model, xs: Tensor, mask: Tensor, surrogate_grand: Tensor, surrogate_null: Tensor):
    xs, mask, tt = _fw_xs_preprocess(xs, mask)
    return model(xs, mask, tt, surrogate_grand, surrogate_null), None


def fw_final(model, xs: Tensor) -> Tuple[Tensor, Tensor]:
    mask = torch.ones_like(xs)
    return model(xs, mask, None)
