"""Mirror of reference models/froyo_bert.py: vanilla BERT modules with the backbone frozen at
``train()`` (:66-103) and a ``Final`` that shares one backbone pass between three heads (:106-204)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn
from typing_extensions import Self

from .. import engine
from ..utils.nnmodel import ObservableModuleMixin, freeze_model_parameters
from .duo_vanilla_bert import _FIELDS
from .vanilla_bert import (VanillaBertClassifier, VanillaBertConfig, VanillaBertExplainer, VanillaBertModel,
                           VanillaBertPooler, _BertExplainerHead, _BertHead)
from .vanilla_vit import _no_autograd


class FroyoBertConfig(VanillaBertConfig):
    def into(self) -> VanillaBertConfig:
        return VanillaBertConfig(**{k: getattr(self, k) for k in _FIELDS})


class FroyoBertClassifier(VanillaBertClassifier):
    def __init__(self, config: FroyoBertConfig):
        super().__init__(config.into())

    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        freeze_model_parameters(self, "bert")
        freeze_model_parameters(self, "bert_pooler")
        freeze_model_parameters(self, "classifier")
        return self


class FroyoBertSurrogate(VanillaBertClassifier):
    def __init__(self, config: FroyoBertConfig):
        super().__init__(config.into())

    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        freeze_model_parameters(self, "bert")
        return self


class FroyoBertExplainer(VanillaBertExplainer):
    def __init__(self, config: FroyoBertConfig):
        super().__init__(config.into())

    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        freeze_model_parameters(self, "bert")
        return self


class FroyoBertFinal(_BertExplainerHead, ObservableModuleMixin, _BertHead):
    def __init__(self, _config: FroyoBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        config = _config.into()
        self.config = config
        self.bert = VanillaBertModel(config)
        self.bert_pooler = VanillaBertPooler(hidden_size=config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self.srg_bert_pooler = VanillaBertPooler(hidden_size=config.hidden_size)
        self.srg_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.srg_classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.srg_act = nn.Softmax(dim=-1)
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self._build_head(config)
        del self.explainer_dropout  # the reference's Final reuses self.dropout (:186); keeps the key set identical

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor] = None):
        _no_autograd(self)
        dtype = engine.get_precision()
        t = input_ids.shape[1]
        hidden, rows, bits = self.bert.run(input_ids, attention_mask, token_type_ids, cls_only=False)
        self.om_record_features(repr_cls=hidden, repr_srg=hidden, repr_exp=hidden)
        cls_logits = self._pool_classify(hidden, rows, t, self.bert_pooler, self.classifier, "cls", True, dtype)
        grand = null = None
        if self.config.explainer_normalize:
            grand = self._pool_classify(hidden, rows, t, self.srg_bert_pooler, self.srg_classifier, "srg", True, dtype)
            null = self.surrogate_null
        phi = self._run_head(hidden, bits, rows, grand, null, self.config, dtype)
        return cls_logits, phi

    def train(self, mode: bool = True) -> Self:
        super().train(mode)
        freeze_model_parameters(self, "bert")
        freeze_model_parameters(self, "bert_pooler")
        freeze_model_parameters(self, "classifier")
        return self
