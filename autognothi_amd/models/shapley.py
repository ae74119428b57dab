"""Drop-in for reference ``models/shapley.py``: the same six public functions with the same signatures, on the HIP kernels.

| here | reference models/shapley.py | kernel |
|---|---|---|
| ``loss_shapley_new``              | :9-53    | ``ag_shapley_loss`` (+ gradient to ``phi``) |
| ``mask_shapley_new``              | :56-79   | ``ag_mask_shapley_new`` (device MT19937, bit-exact) |
| ``normalize_shapley_explanation`` | :82-93   | ``ag_shapley_normalize_rows`` (+ adjoint) |
| ``loss_logits_kl_divergence``     | :96-106  | ``ag_kl_loss`` (+ gradient to ``current``) |
| ``mask_purely_uniform``           | :109-115 | ``ag_mask_purely_uniform`` |
| ``mask_uniform_selective``        | :118-128 | host ``random`` (as the reference: python stdlib shuffles) |

Differences a caller can see: results live on the GPU (the reference returns CPU masks and moves them with ``.to(device)``,
a no-op here); the losses are 0-dim tensors with a ``grad_fn`` exactly like the reference's.

**Generator semantics.**  The reference draws masks from torch's global CPU generator (``torch.rand``), reseeded per epoch
by ``set_iterative_seed`` and shared with every other host-side draw (e.g. the DataLoader's shuffle seed).  By default the
samplers here do the same thing bit for bit: the global CPU generator's MT19937 state is imported into the device
generator, the masks are drawn on the device, and the advanced state is handed back (``GENERATOR = "shared"``: one
2.5 KB upload + one read-back per call).  ``GENERATOR = "device"`` keeps one device-resident stream instead — seeded from
``torch.initial_seed()`` whenever that changes (i.e. after ``torch.manual_seed`` / ``set_iterative_seed``), never
synchronising; identical masks as long as nothing else draws from the CPU generator in between (the pipelines in
``autognothi_amd/scripts`` use that mode through their own ``DeviceMT19937``).
"""
from __future__ import annotations

import os
import random
from typing import List, Optional

import torch
from torch import Tensor

from .. import _lib as L
from .. import ops

GENERATOR = os.environ.get("AG_MASK_GENERATOR", "shared")   # "shared" | "device"
_DEV_GEN = {}


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("autognothi_amd.models.shapley: the samplers run on an MI355X; no GPU is visible (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class _Stream:
    """context: a DeviceMT19937 positioned where the global CPU generator is; on exit the CPU generator is advanced."""

    def __init__(self, device: torch.device):
        self.device = device

    def __enter__(self) -> ops.DeviceMT19937:
        key = str(self.device)
        ent = _DEV_GEN.get(key)
        if ent is None:
            ent = _DEV_GEN[key] = {"rng": ops.DeviceMT19937(self.device), "seed": None}
        self.ent = ent
        if GENERATOR == "shared":
            ent["rng"].import_torch_cpu_state()
        else:
            seed = int(torch.initial_seed())
            if ent["seed"] != seed:
                ent["rng"].seed(seed & 0xFFFFFFFF)
                ent["seed"] = seed
        return ent["rng"]

    def __exit__(self, *exc) -> bool:
        if GENERATOR == "shared" and exc[0] is None:
            self.ent["rng"].export_to_torch_cpu()
        return False


def mask_shapley_new(n_mask_samples: int, n_players: int, device: Optional[torch.device] = None) -> Tensor:
    """reference :56-79 -> int64 [n_mask_samples, n_players] (paired rows 2i / 2i+1 are complements), on the GPU."""
    assert n_mask_samples % 2 == 0
    dev = torch.device(device) if device is not None else _device()
    with _Stream(dev) as rng:
        masks, _ = ops.mask_shapley_new(rng, n_mask_samples, n_players, want_i64=True, want_bits=False)
    return masks


def mask_purely_uniform(batch_size: int, n_features: int, device: Optional[torch.device] = None) -> Tensor:
    """reference :109-115 -> int64 [batch_size, n_features] on the GPU."""
    dev = torch.device(device) if device is not None else _device()
    with _Stream(dev) as rng:
        masks, _ = ops.mask_purely_uniform(rng, batch_size, n_features, want_i64=True, want_bits=False)
    return masks


def mask_uniform_selective(batch_size: int, n_features: int, n_masked: int) -> Tensor:
    """reference :118-128: exactly ``n_masked`` features off per row, chosen by python's ``random.shuffle`` (host stdlib
    generator, as the reference; not a device path and not on the hot loop)."""
    ret: List[List[int]] = []
    for _ in range(batch_size):
        order = list(range(n_features))
        random.shuffle(order)
        off = set(order[:n_masked])
        ret.append([0 if i in off else 1 for i in range(n_features)])
    return torch.tensor(ret, dtype=torch.long)


def _bits(mask: Tensor, n_players: int) -> Tensor:
    """[B,K,P] / [B*K,P] int64 0/1 (or already packed key bits [B*K, Tw] int32) -> key bits."""
    if mask.dtype == torch.int32 and mask.shape[-1] == ops.mask_words(n_players):
        return mask.reshape(-1, mask.shape[-1]).contiguous()
    return ops.pack_mask(mask.reshape(-1, n_players))


class _ShapleyLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, phi, bits, v_0, v_s, batch_size, n_mask_samples):
        loss, dphi = ops.shapley_loss(bits, v_0, v_s, phi, batch_size, n_mask_samples, want_grad=True)
        ctx.save_for_backward(dphi)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (dphi,) = ctx.saved_tensors
        return dphi * g, None, None, None, None, None


def loss_shapley_new(batch_size: int, n_mask_samples: int, n_players: int, mask: Tensor, v_0: Tensor, v_s: Tensor,
                     v_1: Tensor, phi: Tensor) -> Tensor:
    """reference :9-53: ``P * mse(v_0 + mask @ phi^T, v_s)`` -> scalar; differentiable w.r.t. ``phi`` (``v_1`` is accepted
    and unused, as in the reference)."""
    L.require_gpu(mask, v_0, v_s, phi)
    bits = _bits(mask, n_players)
    return _ShapleyLoss.apply(phi, bits, v_0.detach(), v_s.detach(), batch_size, n_mask_samples)


class _NormalizeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, grand, null):
        p = pred.contiguous().float()
        b, t, c = p.shape
        out = torch.empty_like(p)
        with L.on(p.device):
            L.check(L.lib().ag_shapley_normalize_rows(L.ptr(p), L.ptr(grand.contiguous().float()),
                                                      L.ptr(null.reshape(-1).contiguous().float()), b, t, c, L.ptr(out), L.stream()))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        g = g.contiguous().float()
        b, t, c = g.shape
        d = torch.empty_like(g)
        with L.on(g.device):
            L.check(L.lib().ag_shapley_normalize_rows_bwd(L.ptr(g), b, t, c, L.ptr(d), L.stream()))
        return d, None, None


def normalize_shapley_explanation(pred: Tensor, grand: Tensor, null: Tensor) -> Tensor:
    """reference :82-93: pred [B, T, C] -> pred + ((grand - null) - sum_t pred) / T, all T rows kept (the models fuse this
    with the CLS drop + permute: ``ag_shapley_normalize``); differentiable w.r.t. ``pred``."""
    L.require_gpu(pred, grand, null)
    return _NormalizeRows.apply(pred, grand.detach(), null.detach())


class _KLLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, current, ref):
        loss, dcur = ops.kl_loss(ref, current, want_grad=True)
        ctx.save_for_backward(dcur)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (dcur,) = ctx.saved_tensors
        return dcur * g, None


def loss_logits_kl_divergence(ref: Tensor, current: Tensor) -> Tensor:
    """reference :96-106: ``kl_div(log_softmax(ref), softmax(current), "batchmean")`` (both arguments are already
    probabilities in every caller — the quirk is kept); differentiable w.r.t. ``current``."""
    L.require_gpu(ref, current)
    return _KLLoss.apply(current, ref.detach())
