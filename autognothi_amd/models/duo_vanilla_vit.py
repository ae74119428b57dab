"""Mirror of reference models/duo_vanilla_vit.py: one backbone trained for both objectives; the
explainer returns ``(phi, class probabilities)`` from a single backbone pass (:111-134)."""
from __future__ import annotations

from typing import Tuple

import pydantic
import torch
from torch import Tensor, nn

from .. import _lib as L
from .. import autograd as _ag
from .. import engine, ops
from ..utils.nnmodel import ObservableModuleMixin
from .vanilla_vit import (VanillaViTClassifier, VanillaViTConfig, VanillaViTModel, VanillaViTSurrogate, _ExplainerHead,
                          _no_autograd)

_FIELDS = list(VanillaViTConfig.model_fields.keys())


class DuoVanillaViTConfig(VanillaViTConfig):
    """Same fields as Vanilla ViT (reference :19-35)."""

    @property
    def is_decoder(self) -> bool:
        return False

    def into(self) -> VanillaViTConfig:
        return VanillaViTConfig(**{k: getattr(self, k) for k in _FIELDS})


class DuoVanillaViTClassifier(VanillaViTClassifier):
    def __init__(self, config: DuoVanillaViTConfig):
        super().__init__(config.into())


class DuoVanillaViTSurrogate(VanillaViTSurrogate):
    def __init__(self, config: DuoVanillaViTConfig):
        super().__init__(config.into())


class DuoVanillaViTExplainer(_ExplainerHead, ObservableModuleMixin):
    def __init__(self, config: DuoVanillaViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.vit = VanillaViTModel(config.into())
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self._build_head(config)
        self._cls_head = None

    def forward(self, pixel_values: Tensor, attention_mask: Tensor, surrogate_grand: Tensor,
                surrogate_null: Tensor) -> Tuple[Tensor, Tensor]:
        if _ag.grad_mode(self):   # scripts/train_duo_explainer.py:180-198: both outputs carry gradients
            return _ag.explainer_forward(self, pixel_values, attention_mask, surrogate_grand, surrogate_null)
        dtype = engine.get_precision()
        bits = engine.to_mask_bits(attention_mask, self.vit.n_players)
        hidden, rows = self.vit.run(pixel_values, bits, cls_only=False)
        zs, _ = self.vit.final_norm(hidden, rows, False, dtype, want_f32=False)
        t, h = self.vit.n_players + 1, self.config.hidden_size
        z = zs.view(rows, t, h)
        self.om_record_features(repr_cls=z, repr_exp=z)
        if self._cls_head is None:
            self._cls_head = engine.PackedLinear([self.classifier.weight], [self.classifier.bias])
        logits = ops.softmax_rows(engine.linear_head(zs, t * h, rows, self._cls_head, L.AG_EPI_BIAS_F32, dtype))
        phi = self._run_head(z, bits, rows, surrogate_grand, surrogate_null, self.config, dtype)
        return phi, logits


class DuoVanillaViTFinal(nn.Module, ObservableModuleMixin):
    def __init__(self, config: DuoVanillaViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.surrogate = VanillaViTSurrogate(config.into())
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self.explainer = DuoVanillaViTExplainer(config)

    def forward(self, pixel_values: Tensor, attention_mask: Tensor) -> Tuple[Tensor, Tensor]:
        if self.config.explainer_normalize:
            surrogate_grand = self.surrogate(pixel_values, attention_mask)
            om_repr_srg = self.surrogate.om_take_observations()
        else:
            surrogate_grand, om_repr_srg = None, {}
        explainer, logits = self.explainer(pixel_values, attention_mask, surrogate_grand, self.surrogate_null)
        om_repr_exp = self.explainer.om_take_observations()
        self.om_record_features(repr_cls=om_repr_exp.get("repr_cls", None), repr_srg=om_repr_srg.get("repr_srg", None),
                                repr_exp=om_repr_exp.get("repr_exp", None))
        return logits, explainer

    def om_retain_observations(self, flag: bool = True) -> None:
        ObservableModuleMixin.om_retain_observations(self, flag)
        self.surrogate.om_retain_observations(flag)
        self.explainer.om_retain_observations(flag)
