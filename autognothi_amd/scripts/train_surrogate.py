"""Per-batch pieces of reference scripts/train_surrogate.py (:112-211) available on the HIP path:
the uniform mask sampler, the frozen-classifier target forward, the masked surrogate forward and the
KL loss with its gradient w.r.t. the surrogate's output, and the training epoch built on them
(``surrogate_epoch_train``: HIP forward + backward, reference optimiser)."""
from __future__ import annotations

from typing import Any, Callable, Iterable, Optional, Tuple

import torch
from torch import Tensor

from .. import distributed, ops
from ..recipes.types import ModelRecipe
from .common import Log, device_rng


def surrogate_batch_loss(recipe: ModelRecipe, m_classifier, m_surrogate, xs: Tensor, n_players: int, rng):
    """reference :133-149 forward: mask_purely_uniform -> classifier(all ones) -> surrogate(masked) ->
    kl(log_softmax(orig), softmax(adapt)) on the already-softmaxed outputs (the quirk is preserved).
    -> (loss [1], d loss / d adapt_Ys [B,C], orig_Ys, adapt_Ys)."""
    b = xs.shape[0]
    _, bits = ops.mask_purely_uniform(rng, b, n_players, want_i64=False, want_bits=True)
    ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
    with torch.no_grad():
        _, orig = recipe.fw_classifier(m_classifier, xs, ones)   # the SECOND output, as the reference (:141): LTT returns (side, backbone)
        adapt, _ = recipe.fw_surrogate(m_surrogate, xs, bits)
    loss, dcur = ops.kl_loss(orig, adapt)
    return loss, dcur, orig, adapt


def surrogate_epoch_eval(env: Any, device: torch.device, n_players: int, d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe,
                         m_classifier, m_surrogate, epoch: int, gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]],
                         seed: Optional[int] = None) -> float:
    """reference _surrogate_epoch_eval (:163-211) -> mean KL loss."""
    env = env or Log()
    rng = device_rng(m_surrogate, device, seed)
    m_classifier.eval(); m_surrogate.eval()
    tot, n = 0.0, 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, _ = gen_input(_inputs, _targets)
        loss, _, _, _ = surrogate_batch_loss(m_recipe, m_classifier, m_surrogate, xs, n_players, rng)
        tot += float(loss.item()) * xs.shape[0]
        n += xs.shape[0]
        env.log(f"  > epoch {epoch} :{batch_idx}:test // loss: kl {tot / n:.6f}")
    return tot / max(n, 1)


def surrogate_epoch_train(env: Any, device: torch.device, n_players: int, d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe,
                          m_classifier, m_surrogate, optimizer: torch.optim.Optimizer, epoch: int,
                          gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None) -> float:
    """reference _surrogate_epoch_train (:112-160): uniform masks, frozen-classifier targets (no grad), masked
    surrogate forward + KL + backward on the HIP training kernels, reference optimiser step.  -> mean KL."""
    from ..training import make_surrogate_trainer
    env = env or Log()
    rng = device_rng(m_surrogate, device, seed)
    trainer = m_surrogate.__dict__.get("_ag_trainer") or make_surrogate_trainer(m_recipe, m_surrogate)
    m_surrogate.__dict__["_ag_trainer"] = trainer
    m_classifier.eval()
    m_surrogate.train()
    tot, n = 0.0, 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, _ = gen_input(_inputs, _targets)
        b = xs.shape[0]
        optimizer.zero_grad()
        _, bits = ops.mask_purely_uniform(rng, b, n_players, want_i64=False, want_bits=True)
        ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
        with torch.no_grad():
            _, orig = m_recipe.fw_classifier(m_classifier, xs, ones)   # second output (reference :141)
        loss, _probs = trainer.loss_and_grads(xs, bits, orig, train=True, seed=(seed or 0) + epoch)
        distributed.allreduce_grads([p for p in m_surrogate.parameters() if p.requires_grad], average=True)
        optimizer.step()
        tot += float(loss.item()) * b
        n += b
        env.log(f"  > epoch {epoch} :{batch_idx}:train // loss: kl {tot / n:.6f}")
    return tot / max(n, 1)
