"""Mirror of reference models/duo_vanilla_bert.py.  Note the asymmetry the reference has and parity
keeps: the duo-BERT explainer's class head returns RAW logits, no soft-max (:142-144), and its
``forward`` returns ``(logits, phi)`` (:161) which the recipe swaps."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .. import engine
from ..utils.nnmodel import ObservableModuleMixin
from .vanilla_bert import (VanillaBertClassifier, VanillaBertConfig, VanillaBertModel, VanillaBertPooler,
                           VanillaBertSurrogate, _BertExplainerHead, _BertHead)
from .. import autograd as _ag
from .vanilla_vit import _no_autograd

_FIELDS = list(VanillaBertConfig.model_fields.keys())


class DuoVanillaBertConfig(VanillaBertConfig):
    def into(self) -> VanillaBertConfig:
        return VanillaBertConfig(**{k: getattr(self, k) for k in _FIELDS})


class DuoVanillaBertClassifier(VanillaBertClassifier):
    def __init__(self, config: DuoVanillaBertConfig):
        super().__init__(config.into())


class DuoVanillaBertSurrogate(VanillaBertSurrogate):
    def __init__(self, config: DuoVanillaBertConfig):
        super().__init__(config.into())


class DuoVanillaBertExplainer(_BertExplainerHead, ObservableModuleMixin, _BertHead):
    def __init__(self, config: DuoVanillaBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.bert = VanillaBertModel(config.into())
        self.bert_pooler = VanillaBertPooler(hidden_size=config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self._build_head(config)

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor],
                surrogate_grand: Tensor, surrogate_null: Tensor) -> Tuple[Tensor, Tensor]:
        if _ag.grad_mode(self):   # scripts/train_duo_explainer.py:180-198: both outputs carry gradients
            phi, logits = _ag.explainer_forward(self, input_ids, attention_mask, surrogate_grand, surrogate_null)
            return logits, phi
        dtype = engine.get_precision()
        hidden, rows, bits = self.bert.run(input_ids, attention_mask, token_type_ids, cls_only=False)
        self.om_record_features(repr_cls=hidden, repr_exp=hidden)
        logits = self._pool_classify(hidden, rows, input_ids.shape[1], self.bert_pooler, self.classifier, "duo", False, dtype)
        phi = self._run_head(hidden, bits, rows, surrogate_grand, surrogate_null, self.config, dtype)
        return logits, phi


class DuoVanillaBertFinal(nn.Module, ObservableModuleMixin):
    def __init__(self, config: DuoVanillaBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.surrogate = VanillaBertSurrogate(config.into())
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self.explainer = DuoVanillaBertExplainer(config)

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor] = None):
        if self.config.explainer_normalize:
            surrogate_grand = self.surrogate(input_ids, attention_mask, token_type_ids)
            om_repr_srg = self.surrogate.om_take_observations()
        else:
            surrogate_grand, om_repr_srg = None, {}
        logits, explainer = self.explainer(input_ids, attention_mask, token_type_ids, surrogate_grand, self.surrogate_null)
        om_repr_exp = self.explainer.om_take_observations()
        self.om_record_features(repr_cls=om_repr_exp.get("repr_cls", None), repr_srg=om_repr_srg.get("repr_srg", None),
                                repr_exp=om_repr_exp.get("repr_exp", None))
        return logits, explainer

    def om_retain_observations(self, flag: bool = True) -> None:
        ObservableModuleMixin.om_retain_observations(self, flag)
        self.surrogate.om_retain_observations(flag)
        self.explainer.om_retain_observations(flag)
