"""torch.autograd bridge: makes ``recipe.fw_explainer`` / ``recipe.fw_surrogate`` differentiable, so the reference's own
training loops run on the HIP path unchanged.

The reference calls the explainer / surrogate with grad enabled and then ``loss.backward()``
(scripts/train_explainer.py:183-197, scripts/train_duo_explainer.py:180-198, scripts/train_surrogate.py:143-147).  Here a
module ``forward`` under grad mode returns tensors whose ``grad_fn`` is ``TrainerFn``: its forward runs the training
forward of ``autognothi_amd/training.py`` (HIP kernels, activations saved on the device), its backward runs the
``train.hip`` backward from d loss / d output and hands every trainable parameter's gradient back to autograd, which
accumulates it into ``param.grad`` like for any other op (gradient accumulation, hooks and ``zero_grad`` semantics hold).
Under ``torch.no_grad()`` (every inference / measurement caller) the modules take the fast inference path as before.

Dropout seeds follow the global seed (``torch.initial_seed()``, reseeded per epoch by ``set_iterative_seed``) and the
trainer's step counter; the keep decisions are a counter hash, not torch's Philox stream.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _lib as L
from . import engine, ops


def grad_mode(module: nn.Module) -> bool:
    """True when a reference training caller is driving ``module`` (grad enabled and something to train)."""
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


class TrainerFn(torch.autograd.Function):
    """forward: ``run()`` -> tuple of output tensors; backward: ``back(*grads)`` fills the parameters' .grad, which are then
    returned to autograd (existing .grad values are set aside during the call and restored, so accumulation is autograd's)."""

    @staticmethod
    def forward(ctx, run: Callable[[], Tuple[Tensor, ...]], back: Callable[..., None], n_diff: int, *params: Tensor):
        outs = run()
        ctx.back, ctx.params = back, params
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*outs[n_diff:])
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        params: Sequence[Tensor] = ctx.params
        held = [p.grad for p in params]
        for p in params:
            p.grad = None
        try:
            with torch.no_grad():
                ctx.back(*gouts)
            new = [p.grad for p in params]
        finally:
            for p, g in zip(params, held):
                p.grad = g
        return (None, None, None, *new)


def _trainer(module: nn.Module, make: Callable) -> object:
    tr = module.__dict__.get("_ag_trainer")
    if tr is None:
        tr = make(None, module)
        module.__dict__["_ag_trainer"] = tr
    return tr


def _seed() -> int:
    return int(torch.initial_seed()) & 0x7FFFFFFF


def _trainable(module: nn.Module) -> List[Tensor]:
    return [p for p in module.parameters() if p.requires_grad]


def _bits(module: nn.Module, attention_mask: Tensor, n_players: int) -> Tensor:
    return engine.to_mask_bits(attention_mask, n_players)


def _frozen_backbone_probs(module: nn.Module, z_cls: Tensor) -> Tensor:
    """The LTT recipes' second output: the frozen backbone's own prediction from its CLS rows (fp32 [B,H]); no gradient,
    no dropout (training callers ignore it: scripts/train_surrogate.py:144, train_explainer.py:184)."""
    f32 = L.AG_F32
    x = z_cls.contiguous().float()
    if hasattr(module, "bert_pooler"):
        pl = module.bert_pooler.dense
        x = ops.gemm(x, pl.weight.detach().float().contiguous(), pl.bias.detach().float().contiguous(), L.AG_EPI_BIAS_TANH, f32)
    cl = module.classifier
    logits = ops.gemm(x, cl.weight.detach().float().contiguous(), cl.bias.detach().float().contiguous(), L.AG_EPI_BIAS_F32, f32)
    return ops.softmax_rows(logits)


def explainer_forward(module: nn.Module, xs: Tensor, attention_mask: Tensor, surrogate_grand: Optional[Tensor],
                      surrogate_null: Optional[Tensor]) -> Tuple[Tensor, Optional[Tensor]]:
    """fw_explainer under grad: -> (phi [B,C,P] with grad_fn, second output or None).  The second output is the duo
    recipes' ``base_Ys`` (differentiable) or the LTT recipes' frozen-backbone prediction (not differentiable)."""
    from .training import make_explainer_trainer
    L.require_gpu(xs)
    tr = _trainer(module, make_explainer_trainer)
    bits = _bits(module, attention_mask, tr.n_players)
    train = module.training
    ltt = type(tr).__name__.startswith("Ltt")
    g = surrogate_grand.detach() if surrogate_grand is not None else None
    n = surrogate_null.detach() if surrogate_null is not None else None
    x = xs.detach()
    prev = engine.precision_name()

    def run():
        try:
            phi, base = tr.forward_phi(x, n, g, train=train, seed=_seed(), bits=bits)
        finally:
            engine.set_precision(prev)
        if ltt:
            h = module.config.hidden_size
            zc = tr.ladder.z_last.view(x.shape[0], -1, h)[:, 0, :]
            return (phi, _frozen_backbone_probs(module, zc))
        return (phi,) if base is None else (phi, base)

    def back(dphi, dbase=None):
        if dphi is None:
            dphi = torch.zeros_like(outs[0])
        tr.backward_phi(dphi.contiguous().float(), dbase)

    n_diff = 1 if (ltt or not tr.duo) else 2
    outs = TrainerFn.apply(run, back, n_diff, *_trainable(module))
    return outs[0], (outs[1] if len(outs) > 1 else None)


def surrogate_forward(module: nn.Module, xs: Tensor, attention_mask: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    """fw_surrogate under grad: -> (probabilities [B,C] with grad_fn, LTT: frozen-backbone probabilities else None)."""
    from .training import make_surrogate_trainer
    L.require_gpu(xs)
    tr = _trainer(module, make_surrogate_trainer)
    if attention_mask.shape[0] != xs.shape[0]:
        raise ValueError("training forward: one mask row per input row (the K-mask sharing extension is inference-only)")
    bits = _bits(module, attention_mask, tr.n_players)
    train = module.training
    ltt = type(tr).__name__.startswith("Ltt")
    x = xs.detach()
    prev = engine.precision_name()

    def run():
        try:
            probs = tr.forward_probs(x, bits, train=train, seed=_seed())
        finally:
            engine.set_precision(prev)
        if ltt:
            h = module.config.hidden_size
            zc = tr.ladder.z_last.view(x.shape[0], -1, h)[:, 0, :]
            return (probs, _frozen_backbone_probs(module, zc))
        return (probs,)

    def back(dprobs, _unused=None):
        if dprobs is None:
            dprobs = torch.zeros_like(outs[0])
        tr.backward_probs(dprobs)

    outs = TrainerFn.apply(run, back, 1, *_trainable(module))
    return outs[0], (outs[1] if len(outs) > 1 else None)
