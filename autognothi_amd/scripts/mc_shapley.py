"""Monte-Carlo permutation Shapley values of ONE input (reference scripts/preview_text_shapley.py:62-153,
utils/functional.py:6-93): ``reps`` random permutations, each walked as P+1 nested masks through the surrogate, the
marginal contributions of the "sharpened" value averaged per player.

The reference pipelines 16-row batches through ``batched()``.  Here all reps*(P+1) masks of the input are built on
the device and run as K-shared rows of ONE input (embeddings and layer-0 LN/QKV computed once; BERT: masked tokens
pruned after layer 0), ``rows_per_pass`` rows at a time, and the chain differences / scatter / mean are one kernel
(``ag_mc_shapley_reduce``).  Permutations come from ``torch.randperm`` on the host generator exactly as the reference
draws them (or are passed in)."""
from __future__ import annotations

from typing import Any, Optional, Tuple

import torch
from torch import Tensor

from .. import ops
from ..recipes.types import ModelRecipe


def nested_masks(perms: Tensor, device: torch.device) -> Tuple[Tensor, Tensor]:
    """perms [reps, P] (a permutation of 0..P-1 per row) -> (key bits of the reps*(P+1) nested masks, rank [reps, P] int32).
    Row r*(P+1)+i has the first i players of permutation r visible (reference _preprocess :81-91)."""
    perms = perms.to(device=device, dtype=torch.int64)
    reps, p = perms.shape
    rank = torch.empty_like(perms)
    rank.scatter_(1, perms, torch.arange(p, device=device).expand(reps, p))          # rank[r, perms[r, j]] = j
    steps = torch.arange(p + 1, device=device).view(1, p + 1, 1)
    masks = (rank.view(reps, 1, p) < steps).to(torch.int64).view(reps * (p + 1), p)   # index plumbing
    return ops.pack_mask(masks), rank.to(torch.int32)


def get_shap(device: torch.device, m_recipe: ModelRecipe, m_surrogate: Any, xs: Tensor, n_players: int, reps: int,
             perms: Optional[Tensor] = None, rows_per_pass: int = 2048) -> Tuple[Tensor, Tensor, Tensor]:
    """(xs [1, ...]) -> (sv [C, P], v0 [C], vn [C]) — reference _get_shap (:62-132)."""
    if xs.shape[0] != 1:
        raise ValueError("get_shap explains one input at a time (reference :68)")
    if perms is None:
        perms = torch.stack([torch.randperm(n_players) for _ in range(reps)])        # host generator, as the reference (:86)
    if perms.shape != (reps, n_players):
        raise ValueError(f"perms must be [{reps}, {n_players}]")
    bits, rank = nested_masks(perms, device)
    xs = xs.to(device)
    m_surrogate.eval()
    outs = []
    with torch.no_grad():
        for r0 in range(0, bits.shape[0], rows_per_pass):
            logits, _ = m_recipe.fw_surrogate(m_surrogate, xs, bits[r0:r0 + rows_per_pass].contiguous())
            outs.append(logits)
    v = torch.cat(outs, dim=0).view(reps, n_players + 1, -1)
    return ops.mc_shapley_reduce(v, rank)
