"""Mirror of reference models/froyo_vit.py ("frozen yoghurt": frozen backbone, trainable heads).
Train-time modules are the vanilla ones with ``train()`` re-freezing ``vit.*`` (:63-97); ``Final``
shares ONE backbone pass between the classifier head, the surrogate head and the explainer head
(:100-171).  The reference's ``FroyoViTFinal.forward`` requires two positional arguments its own
recipe never passes (latent TypeError); here they default to None, and are ignored
exactly where the reference ignores them (:164-169)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn
from typing_extensions import Self

from .. import _lib as L
from .. import engine, ops
from ..utils.nnmodel import ObservableModuleMixin, freeze_model_parameters
from .duo_vanilla_vit import _FIELDS
from .vanilla_vit import (VanillaViTClassifier, VanillaViTConfig, VanillaViTExplainer, VanillaViTModel, _ExplainerHead,
                          _no_autograd)


class FroyoViTConfig(VanillaViTConfig):
    @property
    def is_decoder(self) -> bool:
        return False

    def into(self) -> VanillaViTConfig:
        return VanillaViTConfig(**{k: getattr(self, k) for k in _FIELDS})


class FroyoViTClassifier(VanillaViTClassifier):
    def __init__(self, config: FroyoViTConfig):
        super().__init__(config.into())

    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        freeze_model_parameters(self, "vit")
        freeze_model_parameters(self, "classifier")
        return self


class FroyoViTSurrogate(VanillaViTClassifier):
    def __init__(self, config: FroyoViTConfig):
        super().__init__(config.into())

    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        freeze_model_parameters(self, "vit")
        return self


class FroyoViTExplainer(VanillaViTExplainer):
    def __init__(self, config: FroyoViTConfig):
        super().__init__(config.into())

    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        freeze_model_parameters(self, "vit")
        return self


class FroyoViTFinal(_ExplainerHead, ObservableModuleMixin):
    def __init__(self, config: FroyoViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.vit = VanillaViTModel(config.into())
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self.srg_classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.srg_act = nn.Softmax(dim=-1)
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self._build_head(config)
        self._heads = None

    def forward(self, x: Tensor, attention_mask: Tensor, surrogate_grand: Optional[Tensor] = None,
                surrogate_null: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        _no_autograd(self)
        dtype = engine.get_precision()
        bits = engine.to_mask_bits(attention_mask, self.vit.n_players)
        hidden, rows = self.vit.run(x, bits, cls_only=False)
        zs, _ = self.vit.final_norm(hidden, rows, False, dtype, want_f32=False)
        t, h = self.vit.n_players + 1, self.config.hidden_size
        z = zs.view(rows, t, h)
        self.om_record_features(repr_cls=z, repr_srg=z, repr_exp=z)
        if self._heads is None:
            self._heads = (engine.PackedLinear([self.classifier.weight], [self.classifier.bias]),
                           engine.PackedLinear([self.srg_classifier.weight], [self.srg_classifier.bias]))
        cls_logits = ops.softmax_rows(engine.linear_head(zs, t * h, rows, self._heads[0], L.AG_EPI_BIAS_F32, dtype))
        grand = null = None
        if self.config.explainer_normalize:
            grand = ops.softmax_rows(engine.linear_head(zs, t * h, rows, self._heads[1], L.AG_EPI_BIAS_F32, dtype))
            null = self.surrogate_null
        phi = self._run_head(z, bits, rows, grand, null, self.config, dtype)
        return cls_logits, phi

    def train(self, mode: bool = True) -> Self:
        super().train(mode)
        freeze_model_parameters(self, "vit")
        freeze_model_parameters(self, "classifier")
        return self
