"""Small helpers shared by the pipeline bodies."""
from __future__ import annotations

from typing import Any, Callable, Optional

import torch


# Dropout keys of row-sharded ranks: trainer.loss_and_grads(seed=...) gets ``seed + epoch + DROPOUT_RANK_STRIDE * lo`` (lo = this
# rank's first input of the global batch), so that rank r's local row i does not share its keep pattern with rank 0's row i.
# One rank (lo = 0): the single-process keys.  The keep decisions are a counter hash, not torch's Philox stream, so an N-rank run
# with dropout is statistically — not bitwise — the single-process run; with p = 0 it is equal to it.
DROPOUT_RANK_STRIDE = 1000003


class Log:
    """Stand-in for the reference's ExpEnv (scripts/env.py:13): only ``.log`` is used by the loop bodies."""

    def __init__(self, sink: Optional[Callable[[str], None]] = None):
        self.sink = sink

    def log(self, msg: str) -> None:
        if self.sink is not None:
            self.sink(msg)


def device_rng(holder: Any, device: torch.device, seed: Optional[int]):
    """One device MT19937 per (holder, device); reseeded when `seed` is given (the per-epoch
    set_iterative_seed of scripts/train_explainer.py:64), else continuing its stream."""
    from .. import ops
    cache = holder.__dict__.setdefault("_ag_rng", {})
    key = str(device)
    if key not in cache:
        cache[key] = ops.DeviceMT19937(device, seed if seed is not None else 0)
    elif seed is not None:
        cache[key].seed(seed)
    return cache[key]


class MaskSource:
    """The mask stream of an epoch, as the row-sharded ranks of a run consume it: ONE generator state, identical on every rank;
    each call names the size of the GLOBAL batch and the input range [lo, hi) this rank owns, returns this rank's key bits and
    advances the state by the WHOLE global call (ag_mask_shapley_new_rows / ag_mask_purely_uniform_rows: the other ranks' draws
    are stepped over with twists only).  The union of the ranks' masks is bit-identical to the single-process reference loop
    on the whole batch (models/shapley.py:56-79, :109-115), and at one rank this IS that loop."""

    def __init__(self, rng):
        self.rng = rng

    def shapley(self, n_inputs_total: int, lo: int, hi: int, k: int, n_players: int):
        from .. import ops
        if lo == 0 and hi == n_inputs_total:
            return ops.mask_shapley_new(self.rng, n_inputs_total * k, n_players, want_i64=False, want_bits=True)[1]
        return ops.mask_shapley_new_rows(self.rng, n_inputs_total * k, lo * k, hi * k, n_players, want_i64=False, want_bits=True)[1]

    def uniform(self, n_inputs_total: int, lo: int, hi: int, n_players: int):
        from .. import ops
        if lo == 0 and hi == n_inputs_total:
            return ops.mask_purely_uniform(self.rng, n_inputs_total, n_players, want_i64=False, want_bits=True)[1]
        return ops.mask_purely_uniform_rows(self.rng, n_inputs_total, lo, hi, n_players, want_i64=False, want_bits=True)[1]


class Span:
    """This rank's part of one global batch of ``n_tot`` inputs x ``k`` masks.
    mode "input": inputs [lo, hi) with all k masks each (the default: layer-0 sharing, normalisation and the loss stay
    rank-local).  mode "mask" — fewer inputs than ranks (the ragged tail of an epoch, BASELINE config 4 at one input per step):
    EVERY input, masks [k_lo, k_hi) of each; lo, hi = 0, n_tot."""
    __slots__ = ("n_tot", "lo", "hi", "mode", "k", "k_lo", "k_hi")

    def __init__(self, n_tot: int, lo: int, hi: int, mode: str = "input", k: int = 1, k_lo: int = 0, k_hi: int = 0):
        self.n_tot, self.lo, self.hi, self.mode, self.k, self.k_lo, self.k_hi = n_tot, lo, hi, mode, k, k_lo, k_hi

    @property
    def by_mask(self) -> bool:
        return self.mode == "mask"

    def astuple(self):
        return (self.n_tot, self.lo, self.hi)


def shard_auto(xs, zs, k: int):
    """``shard`` with the fall-back of SURVEY §8e: a batch with fewer inputs than ranks is split by MASK inside every input
    (every rank keeps all inputs and takes a slice of the k masks of each), so that no rank idles; needs k >= ranks.
    -> (xs_local, zs_local, Span)."""
    from .. import distributed
    n = xs.shape[0]
    _, w = distributed.world()
    if w > 1 and 0 < n < w and k >= w:
        k_lo, k_hi = distributed.shard_range(k)
        return xs, zs, Span(n, 0, n, "mask", k, k_lo, k_hi)
    xs_l, zs_l, n_tot, lo, hi = shard(xs, zs)
    return xs_l, zs_l, Span(n_tot, lo, hi, "input", k)


def mask_source(holder: Any, device: torch.device, seed: Optional[int]) -> MaskSource:
    """the epoch's MaskSource on the (holder, device) generator of ``device_rng`` (reseeded when `seed` is given)."""
    return MaskSource(device_rng(holder, device, seed))


def shard(xs, zs=None):
    """this rank's contiguous slice of a global batch (distributed.shard_range) -> (xs_local, zs_local, n_total, lo, hi).
    One rank: the batch itself."""
    from .. import distributed
    n = xs.shape[0]
    lo, hi = distributed.shard_range(n)
    if lo == 0 and hi == n:
        return xs, zs, n, lo, hi
    return xs[lo:hi], (zs[lo:hi] if zs is not None else None), n, lo, hi
