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


def is_main() -> bool:
    return world()[0] == 0


def barrier() -> None:
    if world()[1] > 1:
        dist.barrier()


def all_agree(flag: bool, device: Optional[torch.device] = None) -> bool:
    """True on every rank iff ``flag`` is true on EVERY rank (MIN all-reduce; one rank: the flag).  For per-rank decisions that change the
    ORDER in which a rank issues collectives — the two-stream training epoch issues the next group's target forward (which may contain an
    all-gather for a batch with fewer inputs than ranks) before this group's gradient exchange: all ranks take that schedule or none does."""
    _, w = world()
    if w == 1:
        return bool(flag)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


class MainOnly:
    """Rank-0-only view of the reference's ExpEnv for the pipeline entry points (scripts/train_*.py): ``log`` / ``metrics`` /
    ``flush_cfg`` act on rank 0 and are no-ops elsewhere; everything else (``config``, ``model_path``, ``d_loader`` ...) reads
    through.  The reference is single-process; with one process per GPU, N ranks appending to one ``.log.txt`` and rewriting
    one ``.hparams.json`` would interleave."""

    def __init__(self, env):
        object.__setattr__(self, "_env", env)
        object.__setattr__(self, "_main", is_main())

    def __getattr__(self, name):
        attr = getattr(self._env, name)
        if name in ("log", "metrics", "flush_cfg") and not self._main:
            return lambda *a, **k: None
        return attr

    def __setattr__(self, name, value):
        setattr(self._env, name, value)


def main_only(env):
    return env if (env is None or world()[1] == 1) else MainOnly(env)


def gather_objects(obj) -> list:
    """every rank's python object, in rank order, on every rank (small host-side results: curves, counts)."""
    _, w = world()
    if w == 1:
        return [obj]
    out = [None] * w
    dist.all_gather_object(out, obj)
    return out


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
    """Inverse exchange of shard_masks_within_inputs: the per-rank [n_inputs*kl_r, ...] row blocks back in the global
    [n_inputs*k, ...] input-major order on every rank.  Each rank writes its block into a zero buffer of the global shape and the
    buffers are summed (one all-reduce of a few KB: v_s is C floats per mask row; x + 0 is exact, so the result carries every
    rank's bits unchanged) — no variable-length gather lists, one collective whatever the split."""
    r, w = world()
    if w == 1:
        return local
    lo, hi = shard_range(k, r, w)
    tail = tuple(local.shape[1:])
    full = torch.zeros((n_inputs, k) + tail, dtype=local.dtype, device=local.device)
    full[:, lo:hi] = local.reshape(n_inputs, hi - lo, *tail)
    dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full.reshape(n_inputs * k, *tail)


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


class GradBucketReducer:
    """Gradient exchange of explainer / surrogate training, overlapped with the backward pass.

    The manual backward of ``autognothi_amd/training.py`` / ``training16.py`` walks the graph from the head down and reports every
    parameter whose gradient is final (``ready(p)``: the trainers call it through ``training.GRAD_SINK``).  Ready gradients fill a
    bucket; a full bucket is PACKED — one ``ag_pack_f32_many`` launch for all its tensors, weighted by this rank's share of the
    global batch on the way — into the bucket's persistent flat buffer, every ``.grad`` becomes a view into that buffer (no
    ``torch.cat``, no copy back after the exchange: the optimiser reads the exchanged values in place), and the buffer's collective
    is issued ASYNCHRONOUSLY on RCCL's stream while the backward kernels of the layers below keep the compute stream busy.
    ``finish()`` (before ``optimizer.step()``) flushes the tail bucket and waits for the collectives.

    Exchange (``mode`` / ``AG_GRAD_EXCHANGE``):
      * ``fp32`` (default): ONE all-reduce per bucket, in place on the flat buffer (RCCL picks its channels over the xGMI mesh);
        exact to the order of the sum.
      * ``rsag``: reduce-scatter + all-gather in place (on a full mesh of 7 point-to-point links per GPU every link carries
        a shard, where a single ring is bound by one link) — two collectives per bucket; RCCL only (gloo: as ``fp32``).
      * ``bf16``: the bucket is packed as bf16, exchanged by all-to-all (each rank receives every rank's piece of ITS shard), summed
        in fp32 on receipt, and the fp32 shard is all-gathered: half the reduce-scatter bytes with no bf16 accumulation; the gradient
        every rank ends with is the fp32 sum of bf16-rounded contributions.
    (None of the three has run on more than one MI355X by this builder: the default is the form with the fewest moving parts.)
    Bucket size: xGMI is point-to-point, a collective step is per-link bound, so buckets are large (default 64 MiB: the vanilla
    ViT-base explainer's 419 MB of fp32 gradients are 7 collectives, the first one in flight after the head + 2 layers).
    Every parameter must be reported at most once per step; ``finish()`` also reduces trainable parameters that were never
    reported but hold a gradient (callers that do not instrument their backward lose the overlap, not the result)."""

    PAD = 8      # elements: every segment of a bucket starts 16-byte aligned in the bf16 form too

    def __init__(self, params: Iterable[Tensor], bucket_bytes: int = 64 << 20, average: bool = True, mode: Optional[str] = None):
        import os
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes, self.average = int(bucket_bytes), average
        self.mode = (mode or os.environ.get("AG_GRAD_EXCHANGE", "fp32")).lower()
        if self.mode not in ("fp32", "bf16", "rsag"):
            raise ValueError(f"GradBucketReducer: unknown exchange mode {self.mode!r} (fp32 | rsag | bf16)")
        # rsag / bf16 have run on gloo (world 2) and on RCCL at world size 1 only: on RCCL with more than one rank they need
        # AG_GRAD_EXCHANGE_UNPROVEN=1 beside the mode until tests/test_gpu_rccl.py has covered them on two GPUs (no such box was available
        # to the builder); without it the reducer falls back to the one in-place all-reduce per bucket
        self._unproven_ok = os.environ.get("AG_GRAD_EXCHANGE_UNPROVEN", "0") == "1"
        self._pending: List[Tensor] = []
        self._pending_bytes = 0
        self._inflight: List[Tuple] = []
        self._seen = set()
        self.collectives = 0
        self._weight: Optional[float] = None
        self._known = None            # ids of the parameters that received a gradient in the last regular (non-ragged) step
        self._bufs = {}               # bucket index of the step -> (flat fp32 buffer, bf16 send / receive buffers or None)
        self._layouts = {}            # bucket index -> the bucket's cached layout (see _flush)

    def begin(self, weight: Optional[float] = None) -> None:
        """Start a step.  ``weight`` = this rank's share of the step's global batch (inputs of this rank / inputs of all ranks):
        the exchange then computes sum_r weight_r * grad_r — the gradient of the GLOBAL batch-mean loss when every rank
        back-propagated its LOCAL batch-mean loss, also for ragged shards — instead of the plain average (weight None)."""
        self._weight = weight

    def ready(self, p: Tensor) -> None:
        _, w = world()
        if w == 1 or p.grad is None:
            return
        if id(p) in self._seen:
            raise RuntimeError("GradBucketReducer.ready: a parameter was reported twice in one step")
        self._seen.add(id(p))
        self._pending.append(p)
        self._pending_bytes += p.grad.numel() * 4
        if self._pending_bytes >= self.bucket_bytes:
            self._flush()

    # ---- one bucket: pack -> views -> collective
    def _buffers(self, index: int, n: int, device, bf16: bool, w: int):
        have = self._bufs.get(index)
        if have is None or have[0].numel() < n or have[0].device != device or (bf16 and have[1] is None):
            # (zeroed once: the padding between segments and behind the last one takes part in every sum)
            flat = torch.zeros(n, dtype=torch.float32, device=device)
            send = torch.zeros(n, dtype=torch.bfloat16, device=device) if bf16 else None
            recv = torch.zeros(n, dtype=torch.bfloat16, device=device) if bf16 else None
            have = (flat, send, recv)
            self._bufs[index] = have
        return have

    def _flush(self) -> None:
        if not self._pending:
            return
        w = dist.get_world_size()                                # (the group's real size: shapes of the collectives)
        ps = self._pending
        pad = self.PAD
        dev = ps[0].grad.device
        mode = self.mode
        if mode != "fp32" and w > 1 and dist.get_backend() == "nccl" and not self._unproven_ok:
            mode = "fp32"
        bf16 = mode == "bf16"
        # the layout of a bucket — offsets, the views that become .grad, the destinations of the pack — is the same step after step (the
        # backward reports the same parameters in the same order): built once per bucket, reused while the reported sequence matches
        key = tuple((id(q), tuple(q.grad.shape), str(q.grad.device)) for q in ps)
        lay = self._layouts.get(self.collectives)
        if lay is None or lay[0] != key or lay[1] != (str(dev), bf16, w):
            offs, n = [], 0
            for q in ps:
                offs.append(n)
                n += -(-q.grad.numel() // pad) * pad
            n = -(-n // (pad * w)) * (pad * w)                   # equal, aligned shards
            flat, send, recv = self._buffers(self.collectives, n, dev, bf16, w)
            flat, send, recv = flat[:n], (send[:n] if bf16 else None), (recv[:n] if bf16 else None)
            target = send if bf16 else flat
            views = [flat[o:o + q.grad.numel()].view_as(q.grad) for q, o in zip(ps, offs)]
            dsts = [target[o:o + q.grad.numel()] for q, o in zip(ps, offs)]
            lay = (key, (str(dev), bf16, w), offs, n, flat, send, recv, views, dsts)
            self._layouts[self.collectives] = lay
        _, _, offs, n, flat, send, recv, views, dsts = lay
        scale = 1.0 if self._weight is None else float(self._weight)
        target = send if bf16 else flat
        grads = [q.grad if (q.grad.dtype == torch.float32 and q.grad.is_contiguous()) else q.grad.float().contiguous() for q in ps]
        if dev.type == "cuda":
            from . import ops
            pairs = [(g, d_) for g, d_, v in zip(grads, dsts, views) if bf16 or g.data_ptr() != v.data_ptr() or scale != 1.0]
            ops.pack_many(pairs, scale)                          # ONE launch (per 96 tensors) for the whole bucket
        elif bf16:                                               # (CPU / gloo: the sharding contract's tests)
            for g, o in zip(grads, offs):
                target[o:o + g.numel()].copy_((g.reshape(-1) * scale).to(torch.bfloat16))
        else:
            for g, v in zip(grads, views):
                if g.data_ptr() != v.data_ptr():
                    v.copy_(g)
                if scale != 1.0:
                    v.mul_(scale)
        for q, v in zip(ps, views):                              # the gradient IS the bucket's slice from here on
            q.grad = v
        backend = dist.get_backend()
        if bf16 and dev.type != "cuda":
            # (gloo: the same exchange, blocking — pieces, shard and gather indices as on RCCL)
            send.view(w, n // w)             # (n is a multiple of w: equal pieces)
            if w > 1:
                dist.all_to_all_single(recv, send)
            else:
                recv.copy_(send)
            r = dist.get_rank()
            shard = flat[r * (n // w):(r + 1) * (n // w)]
            shard.copy_(recv.view(w, n // w).float().sum(dim=0))
            work = dist.all_gather_into_tensor(flat, shard.clone(), async_op=True)
            self._inflight.append((work, flat, None))
        elif bf16:
            # all-to-all of bf16 pieces, fp32 sum on receipt, all-gather of the fp32 shard — behind the a2a on a side stream, so that the
            # backward's stream never waits for a collective
            side = self._side_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                dist.all_to_all_single(recv, send)
                r = dist.get_rank()
                shard = flat[r * (n // w):(r + 1) * (n // w)]
                torch.sum(recv.view(w, n // w).float(), dim=0, out=shard)
                work = dist.all_gather_into_tensor(flat, shard, async_op=True)
            for t_ in (flat, send, recv):
                t_.record_stream(side)
            self._inflight.append((work, flat, side))
        elif mode == "rsag" and backend == "nccl":
            r = dist.get_rank()
            shard = flat[r * (n // w):(r + 1) * (n // w)]
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, async_op=True)
            work = dist.all_gather_into_tensor(flat, shard, async_op=True)     # (same communicator stream: ordered behind the reduce-scatter)
            self._inflight.append((work, flat, None))
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            self._inflight.append((work, flat, None))
        self._pending, self._pending_bytes = [], 0
        self.collectives += 1

    _SIDE = {}

    @classmethod
    def _side_stream(cls, dev):
        key = str(dev)
        if key not in cls._SIDE:
            cls._SIDE[key] = torch.cuda.Stream(device=dev)
        return cls._SIDE[key]

    def finish(self, fill_missing: bool = False) -> int:
        """-> number of collectives of this step.  ``fill_missing``: trainable parameters without a gradient take part as
        zeros (a rank whose shard of the step is EMPTY still has to enter every collective, with the same bucket layout as
        the others: such steps run un-instrumented on all ranks, in parameter order)."""
        _, w = world()
        if w > 1:
            if not fill_missing:                  # a regular step: remember which parameters a backward gives a gradient to
                self._known = {id(p) for p in self.params if p.grad is not None}
            for p in self.params:                 # gradients nobody reported (un-instrumented backward)
                if p.grad is None and fill_missing and (self._known is None or id(p) in self._known):
                    # (only parameters a backward DOES reach: a zero gradient for, e.g., a head outside the loss would make AdamW
                    # decay it and start its moments on ragged steps only — the single-process run leaves it untouched)
                    p.grad = torch.zeros_like(p, dtype=torch.float32)
                if p.grad is not None and id(p) not in self._seen:
                    self.ready(p)
            self._flush()
            for work, flat, side in self._inflight:
                work.wait()
                if side is not None:
                    torch.cuda.current_stream(flat.device).wait_stream(side)
                if self.average and self._weight is None:
                    flat /= w
        n_coll = self.collectives
        self._inflight, self._seen, self.collectives, self._weight = [], set(), 0, None
        return n_coll


class ShardedMaskStream:
    """The ONE mask stream of the unsharded loop, consumed by row-sharded ranks without any traffic: every rank seeds the same
    device generator; per step each rank asks for ITS rows of the global call ``mask_shapley_new(n_inputs_total * k, P)``
    (ag_mask_shapley_new_rows: the other ranks' draws are stepped over, twists only, and the state advances by the whole
    call).  The union of the ranks' masks — and everything derived from them — is bit-identical to a single-process run on
    the concatenated batch (row order ``[b0 s0, b0 s1, b1 s0, ...]``, models/shapley.py:24; k is even, so a complementary
    row pair never straddles two inputs)."""

    def __init__(self, device: torch.device, seed: int):
        from . import ops
        self.ops = ops
        self.rng = ops.DeviceMT19937(device, seed)

    def seed(self, seed: int) -> "ShardedMaskStream":
        self.rng.seed(seed)
        return self

    def sample(self, n_inputs_total: int, lo: int, hi: int, k: int, n_players: int, want_i64: bool = False):
        """-> (int64 masks or None, key bits) of inputs [lo, hi) out of the n_inputs_total inputs of this step."""
        return self.ops.mask_shapley_new_rows(self.rng, n_inputs_total * k, lo * k, hi * k, n_players, want_i64=want_i64,
                                              want_bits=True)


def reduce_scalars(values: Sequence[float], device: torch.device) -> List[float]:
    r, w = world()
    if w == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
