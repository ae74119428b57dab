"""Per-batch bodies of reference scripts/train_explainer.py on the HIP path.

``surrogate_targets``     = the hot K-mask loop (:149-179): device mask sampler + B*K masked surrogate forwards
                            (inputs shared across the K masks) + the all-ones "grand" forward.
``explainer_epoch_eval``  = :210-281 (no grad): targets + explainer forward + Shapley loss.
``explainer_batch_loss``  = :184-196 forward part; returns the loss AND d loss / d phi from the HIP loss kernel.

The optimiser step of :197-198 needs the explainer's backward through the transformer, which this
round's kernel set does not contain yet; ``explainer_epoch_train`` therefore raises instead of
falling back to eager PyTorch.
"""
from __future__ import annotations

from typing import Any, Callable, Iterable, Optional, Tuple

import torch
from torch import Tensor

from .. import ops
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


def explainer_epoch_train(*args, **kwargs):
    raise NotImplementedError(
        "explainer training needs the backward kernels of the masked transformer, which are not built in this "
        "round; the hot K-mask target loop (surrogate_targets) and the loss/gradient kernel (explainer_batch_loss) "
        "are available. There is deliberately no eager-PyTorch fallback.")
