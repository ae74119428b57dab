"""Per-batch bodies of reference scripts/train_explainer.py on the HIP path.

``surrogate_targets``     = the hot K-mask loop (:149-179): device mask sampler + B*K masked surrogate forwards
                            (inputs shared across the K masks) + the all-ones "grand" forward.
``explainer_epoch_eval``  = :210-281 (no grad): targets + explainer forward + Shapley loss.
``explainer_batch_loss``  = :184-196 forward part; returns the loss AND d loss / d phi from the HIP loss kernel.

``explainer_epoch_train`` = :128-207: the same targets, then explainer forward + loss + backward on the HIP
training kernels (``autognothi_amd/training.py``) and the reference's own ``torch.optim`` step.
"""
from __future__ import annotations

from typing import Any, Callable, Iterable, Optional, Tuple

import torch
from torch import Tensor

from .. import distributed, ops
from ..recipes.types import ModelRecipe
from .common import Log, device_rng


def surrogate_null(recipe: ModelRecipe, cfg, misc, m_surrogate, device: torch.device) -> Tensor:
    """reference scripts/train_explainer.py:55-60."""
    n_players = recipe.n_players(cfg)
    null_xs = recipe.gen_null(cfg, misc, device)
    m_surrogate.eval()
    with torch.no_grad():
        v0, _ = recipe.fw_surrogate(m_surrogate, null_xs, torch.ones((1, n_players), dtype=torch.long, device=device))
    return v0


def surrogate_targets(recipe: ModelRecipe, m_surrogate, xs: Tensor, n_mask_samples: int, n_players: int, rng) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (mask key bits [B*K, Tw], v_s [B*K, C], v_1 [B, C]); row order [b0 s0, b0 s1, b1 s0, ...]."""
    b = xs.shape[0]
    _, bits = ops.mask_shapley_new(rng, b * n_mask_samples, n_players, want_i64=False, want_bits=True)
    m_surrogate.eval()
    with torch.no_grad():
        v_s, _ = recipe.fw_surrogate(m_surrogate, xs, bits)          # B inputs, B*K mask rows: shared layer 0
        ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
        v_1, _ = recipe.fw_surrogate(m_surrogate, xs, ones)
    return bits, v_s, v_1


def explainer_batch_loss(recipe: ModelRecipe, m_explainer, xs: Tensor, bits: Tensor, v_0: Tensor, v_s: Tensor, v_1: Tensor,
                         n_mask_samples: int, n_players: int, want_grad: bool = False):
    """reference :184-196 (forward): -> (loss [1] device tensor, phi [B,C,P], dphi or None, logits or None)."""
    b = xs.shape[0]
    ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
    with torch.no_grad():
        phi, logits = recipe.fw_explainer(m_explainer, xs, ones, v_1, v_0)
    loss, dphi = ops.shapley_loss(bits, v_0, v_s, phi, b, n_mask_samples, want_grad=want_grad)
    return loss, phi, dphi, logits


def explainer_epoch_eval(env: Any, device: torch.device, n_mask_samples: int, n_players: int, v_0: Tensor,
                         d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, m_explainer, epoch: int,
                         gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None) -> float:
    """reference _explainer_epoch_eval (:210-281) -> test_reg_loss (mean over samples)."""
    env = env or Log()
    rng = device_rng(m_surrogate, device, seed)
    reg_loss, total = 0.0, 0
    m_explainer.eval()
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, _zs = gen_input(_inputs, _targets)
        bits, v_s, v_1 = surrogate_targets(m_recipe, m_surrogate, xs, n_mask_samples, n_players, rng)
        loss, _, _, _ = explainer_batch_loss(m_recipe, m_explainer, xs, bits, v_0, v_s, v_1, n_mask_samples, n_players)
        lv = float(loss.item())
        reg_loss += lv
        total += xs.shape[0]
        env.log(f"  > epoch {epoch} :{batch_idx}:test // loss: shap {lv / xs.shape[0]:.6f}, fin {total}")
    return reg_loss / max(total, 1)


def explainer_epoch_train(env: Any, device: torch.device, n_mask_samples: int, n_players: int, v_0: Tensor,
                          d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, m_explainer,
                          optimizer: torch.optim.Optimizer, epoch: int,
                          gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None) -> float:
    """reference _explainer_epoch_train (:128-207) / _duo_explainer_epoch_train: per batch — K-mask surrogate
    targets (no grad, HIP inference path), explainer forward + Shapley loss + backward (HIP training kernels,
    autognothi_amd/training.py), then the reference's own optimiser step.  -> train_reg_loss (mean)."""
    from ..training import make_explainer_trainer
    env = env or Log()
    rng = device_rng(m_surrogate, device, seed)
    trainer = m_explainer.__dict__.get("_ag_trainer") or make_explainer_trainer(m_recipe, m_explainer)
    m_explainer.__dict__["_ag_trainer"] = trainer
    reg_loss, total = 0.0, 0
    m_explainer.train()
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, zs = gen_input(_inputs, _targets)
        optimizer.zero_grad()
        bits, v_s, v_1 = surrogate_targets(m_recipe, m_surrogate, xs, n_mask_samples, n_players, rng)
        loss, _phi = trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, n_mask_samples, labels=zs, train=True,
                                            seed=(seed or 0) + epoch)
        # N>1 ranks (rows sharded by input): average gradients over RCCL in a few large buckets (no-op at N=1)
        distributed.allreduce_grads([p for p in m_explainer.parameters() if p.requires_grad], average=True)
        optimizer.step()
        lv = float(loss.item())
        reg_loss += lv
        total += xs.shape[0]
        env.log(f"  > epoch {epoch} :{batch_idx}:train // loss: shap {lv / xs.shape[0]:.6f}, fin {total}")
    return reg_loss / max(total, 1)
