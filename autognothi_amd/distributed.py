"""Multi-GPU sharding of the (batch x K-masks) row axis — one process per GPU, torch.distributed over
RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).

The path shards by *input*: rank r owns a contiguous slice of the B inputs together with all K of
their masks, so layer-0 sharing, normalisation and the Shapley loss stay rank-local and the masked
forward needs NO data-path collective.  Exchange steps exist only where the reference's
math couples rows:
  * gather_rows      — all-gather of v_s / phi when a caller wants the global tensor (tiny, latency-bound);
  * allreduce_grads  — explainer training: sum of gradients over ranks, bucketed so each bucket is one large
                       RCCL call (xGMI is 7 point-to-point links per GPU: few large collectives, not many small);
  * reduce_scalars   — loss / count accumulators;
  * shard_masks_within_inputs / gather_masks_within_inputs — fewer inputs than ranks: the K masks of every input are split
                       across ranks instead (reductions over k are additive).
Masks never travel: every rank seeds an identical device generator and slices (bit-exact, cheap).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n_items for `rank`; the first n % world ranks get one extra item."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(masks: Tensor, n_inputs: int, k: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> Tuple[Tensor, int, int]:
    """masks [n_inputs*k, ...] in the reference's [b0 s0, b0 s1, b1 s0, ...] order -> this rank's rows
    (all k masks of its inputs) and its input range."""
    lo, hi = shard_range(n_inputs, rank, world_size)
    return masks[lo * k:hi * k], lo, hi


def shard_masks_within_inputs(masks: Tensor, n_inputs: int, k: int, rank: Optional[int] = None,
                              world_size: Optional[int] = None) -> Tuple[Tensor, int, int]:
    """Fewer inputs than ranks (C4 with one input per step): every rank takes a contiguous slice [klo, khi) of the
    K masks of EVERY input -> (rows [n_inputs*(khi-klo), ...] in the reference's input-major order, klo, khi).  Each rank then
    embeds every input (layer-0 sharing still holds inside a rank); row results are rank-local, and the reductions over k
    (the Shapley loss, sums of v_s) are additive: all-reduce the partial sums (reduce_scalars / allreduce)."""
    klo, khi = shard_range(k, rank, world_size)
    m = masks.reshape(n_inputs, k, *masks.shape[1:])[:, klo:khi]
    return m.reshape(n_inputs * (khi - klo), *masks.shape[1:]).contiguous(), klo, khi


def gather_masks_within_inputs(local: Tensor, n_inputs: int, k: int) -> Tensor:
    """Inverse exchange of shard_masks_within_inputs: all-gather the per-rank [n_inputs*kl_r, ...] row blocks and interleave
    them back into the global [n_inputs*k, ...] input-major order."""
    r, w = world()
    if w == 1:
        return local
    spans = [shard_range(k, q, w) for q in range(w)]
    width = max(hi - lo for lo, hi in spans)
    tail = tuple(local.shape[1:])
    pad = torch.zeros((n_inputs, width) + tail, dtype=local.dtype, device=local.device)
    lo, hi = spans[r]
    pad[:, :hi - lo] = local.reshape(n_inputs, hi - lo, *tail)
    out = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(out, pad)
    return torch.cat([o[:, :h - l] for o, (l, h) in zip(out, spans)], dim=1).reshape(n_inputs * k, *tail)


def shard_auto(masks: Tensor, n_inputs: int, k: int):
    """by input when there is at least one input per rank, else by mask inside every input -> (rows, mode, lo, hi) with mode
    "input" (lo, hi = input range) or "mask" (lo, hi = mask range of every input)."""
    _, w = world()
    if n_inputs >= w:
        rows, lo, hi = shard_rows(masks, n_inputs, k)
        return rows, "input", lo, hi
    rows, lo, hi = shard_masks_within_inputs(masks, n_inputs, k)
    return rows, "mask", lo, hi


def gather_rows(local: Tensor, counts: Sequence[int]) -> Tensor:
    """All-gather variable-length row blocks (counts[r] rows on rank r) into the global tensor."""
    r, w = world()
    if w == 1:
        return local
    width = int(max(counts))
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def allreduce_grads(params: Iterable[Tensor], average: bool = True, bucket_bytes: int = 256 << 20) -> int:
    """Sum (or average) .grad over ranks in flat buckets; returns the number of collectives issued.
    256 MiB buckets: the vanilla ViT-base explainer (104.7 M fp32 grads = 419 MB) is two calls."""
    r, w = world()
    grads = [p.grad for p in params if p.grad is not None]
    if w == 1 or not grads:
        return 0
    calls = 0
    bucket: List[Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size, calls
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= w
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket, size = [], 0
        calls += 1

    for g in grads:
        nbytes = g.numel() * g.element_size()
        if bucket and (size + nbytes > bucket_bytes or bucket[0].dtype != g.dtype):
            flush()
        bucket.append(g)
        size += nbytes
    flush()
    return calls


def reduce_scalars(values: Sequence[float], device: torch.device) -> List[float]:
    r, w = world()
    if w == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
