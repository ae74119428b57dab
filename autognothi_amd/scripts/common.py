"""Small helpers shared by the pipeline bodies."""
from __future__ import annotations

import os
from typing import Any, Callable, Iterable, Optional

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


# ---------------------------------------------------------------------------------------------- the epoch's own stream
_EPOCH_STREAMS = {}


def on_epoch_stream(fn):
    """Run an epoch body (``fn(env, device, ...)``) on a NON-BLOCKING stream of the package's own when the caller's current stream is the
    device's default stream.  torch's default stream is HIP's null stream, and every launch on the null stream is ordered against every
    BLOCKING stream of the process: with RCCL initialised (its streams) or a host program's own streams around, an epoch of 400 launches per
    step on the null stream runs 6-7 % slower (measured, round 5, tools/epoch_stream_probe.py: vanilla ViT-base 8 images x 32 masks, two /
    one stream: 588 / 514 images/s with RCCL up against 623 / 544 on a stream of its own = the rates without RCCL).  The epoch's stream waits
    for the caller's at entry and the caller's for the epoch's at exit: callers see the reference's semantics (scripts/train_explainer.py:128-207
    runs on whatever stream is current).  AG_EPOCH_STREAM=0: the caller's stream as it is."""
    import functools

    import inspect
    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        # (the reference calls its epoch bodies by keyword as well: _explainer_epoch_train(env=..., device=...))
        device = sig.bind(*args, **kwargs).arguments.get("device")
        if getattr(device, "type", None) != "cuda" or os.environ.get("AG_EPOCH_STREAM", "1") == "0":
            return fn(*args, **kwargs)
        cur = torch.cuda.current_stream(device)
        if cur != torch.cuda.default_stream(device) or torch.cuda.is_current_stream_capturing():
            return fn(*args, **kwargs)
        key = str(device)
        st = _EPOCH_STREAMS.get(key)
        if st is None:
            st = _EPOCH_STREAMS[key] = torch.cuda.Stream(device)
        st.wait_stream(cur)
        try:
            with torch.cuda.stream(st):
                return fn(*args, **kwargs)
        finally:
            cur.wait_stream(st)
    return wrapper


# ---------------------------------------------------------------------------------------------- two-stream training epoch (opt-in)
class TrainPartition:
    """The second stream of an explainer training epoch.  The reference runs the K-mask target forward and the explainer's own step
    back to back (scripts/train_explainer.py:153-198); the surrogate is frozen, so the targets of group g + 1 depend on nothing the
    steps of group g do, and the two want different things from the chip: the target forward is the hot path (feed-bound: it loses
    6-13 % on 7/8 - 3/4 of the CUs, throughput ~ CUs^0.43), the step is 400 launches on 1.5 k token rows that fill a quarter of the CUs
    at best.  ``fwd`` is the device's background stream (``_lib.background_stream``: an ordinary non-blocking stream that took its
    hardware queue at the library's first launch — what makes the two streams run beside, not behind, each other) on which the
    persistent large-M GEMM is told to launch only ``n_fwd`` workgroups (ag_set_stream_cus: one per CU, an equal number on every XCD
    and shader engine), so the other CUs are free for the step's kernels on the caller's stream at any moment.  No CU masks:
    hipExtStreamCreateWithCUMask streams are blocking streams — every null-stream operation of the process becomes a barrier across
    them — and buy less (+14 % against +18 %; profiles/HISTORY.md §10)."""

    def __init__(self, device: torch.device, cus_per_xcd_fwd: int):
        from .. import _lib as L
        self.device = device
        self.fwd = L.background_stream(device.index if device.index is not None else torch.cuda.current_device())
        self.n_fwd = 8 * int(cus_per_xcd_fwd)
        self._checked = {}          # raw handle of a caller's stream -> does ``fwd`` run beside it?

    def arm(self) -> Optional["TrainPartition"]:
        """(the background stream is shared by every partition of the device: tell the library THIS one's CU count.)  Before an epoch
        relies on it, the second stream is CHECKED against the caller's stream (``_lib.streams_overlap``: one idle wave on each — one
        kernel's time = beside, two = behind): HIP multiplexes the streams of a process onto a few hardware queues, and a host program that
        created streams / captured graphs before this package first ran can leave the two on ONE queue, where the two-stream epoch is 24 %
        slower than one stream (profiles/HISTORY.md §10).  Then a few fresh streams are tried (each takes the next queue at its first
        submission); None when none runs beside the caller's: the epoch then runs on one stream."""
        from .. import _lib as L
        main = torch.cuda.current_stream(self.device)
        ok = self._checked.get(main.cuda_stream)
        if ok is None:
            ok = L.streams_overlap(main, self.fwd)
            tries = 0
            while not ok and tries < 6:
                cand = torch.cuda.Stream(self.device, priority=-1)
                ok = L.streams_overlap(main, cand)
                if ok:
                    self.fwd = cand
                tries += 1
            self._checked[main.cuda_stream] = ok
        if not ok:
            return None
        with torch.cuda.device(self.device):
            L.check(L.lib().ag_set_stream_cus(self.fwd.cuda_stream, self.n_fwd))
        return self


_PARTITIONS = {}


def train_partition(device: torch.device, m_explainer) -> Optional[TrainPartition]:
    """The epoch's second stream, or None.  ``AG_TRAIN_PARTITION``: "0" = off (targets and steps back to back on the caller's stream); an
    integer = CUs per XCD the target forward's persistent GEMM may take (a multiple of the 4 shader engines); "auto" (default): 24 of 32
    for a ViT explainer whose backbone trains, 16 for a BERT explainer whose backbone trains, 28 for a frozen backbone.  Every rank count: with N > 1 ranks RCCL's
    kernels (the gradient exchange, on the communicator's own stream) are a third client of the CUs the target forward leaves free.  Never
    under the hipGraph step.  None also when the host program used the GPU before it imported this package (``_lib.HIP_TOUCHED_BEFORE_IMPORT``: the one
    state in which the schedule was measured to LOSE; "auto" only) or when the second stream shares the caller's hardware queue (``TrainPartition.arm``)."""
    from .. import _lib as L
    from .. import training16
    mode = os.environ.get("AG_TRAIN_PARTITION", "auto")
    if mode in ("0", "") or device.type != "cuda" or training16.GRAPH_STEP:
        return None
    if mode == "auto" and L.HIP_TOUCHED_BEFORE_IMPORT:
        # the host program used the GPU before importing this package: its streams / graphs took their hardware queues first, the state in which
        # the second stream was measured to run BEHIND the step (-24 … -27 % against one stream); only an explicit AG_TRAIN_PARTITION=<n> opts in
        return None
    if mode == "auto":
        # measured (tools/train_step_bench.py and inside bench.py, 36-72 steps of 8 images x 32 masks, images/s off -> on):
        # vanilla ViT-base 553 -> 638 at 24 (601 at 28, 608 at 20); froyo ViT-base 726 -> 790 at 28 (748 at 24, 688 at 20); duo BERT-base 920 -> 950-1 015 at 28
        # round 6 (the dW products grouped: the step's launches are larger and want more CUs; tools/r6_part_sweep.sh, 36 steps): duo BERT-base 1 051 at 28,
        # 1 090 at 24, 1 146 at 20, **1 172 at 16**, 1 070 at 12; vanilla ViT-base 667 at 24 (630 at 20 / 28); froyo 810 at 28 (764 at 24)
        vit, bert = getattr(m_explainer, "vit", None), getattr(m_explainer, "bert", None)
        if vit is not None and any(q.requires_grad for q in vit.parameters()):
            c = 24
        elif bert is not None and any(q.requires_grad for q in bert.parameters()):
            c = 16
        else:
            c = 28
    else:
        c = int(mode)
    props = torch.cuda.get_device_properties(device)
    if props.multi_processor_count % 32 != 0 or not (0 < c < props.multi_processor_count // 8) or c % 4 != 0:
        return None
    key = (str(device), c)
    if key not in _PARTITIONS:
        _PARTITIONS[key] = TrainPartition(device, c)
    return _PARTITIONS[key].arm()


def train_partition_all_ranks(device: torch.device, m_explainer) -> Optional[TrainPartition]:
    """``train_partition`` as ONE decision of all ranks.  Each rank's verdict depends on its own process history (import order) and on a
    wall-clock probe of its own streams, and the two-stream epoch changes the ORDER of a rank's collectives — ``pipelined_targets`` issues the
    targets of group g + 1, which for a batch with fewer inputs than ranks contain an all-gather (``gather_masks_within_inputs``), before the
    gradient exchange of group g.  A rank on the two-stream schedule beside a rank on one stream would mismatch collectives on the one
    communicator: every rank takes the schedule only if every rank can (MIN all-reduce of the local verdict; one rank: the local verdict)."""
    from .. import distributed
    part = train_partition(device, m_explainer)
    if distributed.world()[1] > 1 and not distributed.all_agree(part is not None, device):
        return None
    return part


def log_schedule(env, part: Optional[TrainPartition]) -> None:
    """which schedule this epoch takes (the default depends on import order): always to the ``autognothi_amd.schedule``
    logger (INFO), and into the epoch log when ``env.log_schedule`` is set (the reference's log has no such line: off by default)."""
    import logging
    msg = ("  > schedule: " + (f"two streams (target forward on {part.n_fwd // 8} CUs per XCD of the background stream)" if part is not None
                               else "one stream"))
    logging.getLogger("autognothi_amd.schedule").info(msg)
    if env is not None and getattr(env, "log_schedule", False):
        env.log(msg)


def _cuda_tensors(obj):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _cuda_tensors(o)


def pipelined_targets(groups: Iterable, compute: Callable[[Any], Any], part: Optional[TrainPartition]):
    """yield (group, compute(group)) for every group; the caller runs the group's steps on its own stream.  With ``part`` the targets of
    group g + 1 are issued on ``part.fwd`` BEFORE the caller issues the steps of group g, so the two run side by side (the first
    group's targets have nothing to run beside: they are computed on the caller's stream, with every CU).  Events order producers
    and consumers, ``record_stream`` tells the caching allocator about the second stream of every tensor that crosses.  Masks are
    still drawn group by group, batch by batch, from the one generator: the same masks, targets and steps as without ``part``
    (tests/test_gpu_scripts.py: bit-identical parameters after two epochs)."""
    if part is None:
        for g in groups:
            yield g, compute(g)
        return
    main = torch.cuda.current_stream()

    def launch(g):
        ready = torch.cuda.Event()
        ready.record(main)                                   # the group's inputs were made on the caller's stream
        with torch.cuda.stream(part.fwd):
            part.fwd.wait_event(ready)
            for t in _cuda_tensors(list(g)):
                t.record_stream(part.fwd)
            tg = compute(g)
            done = torch.cuda.Event()
            done.record(part.fwd)
        for t in _cuda_tensors(tg):
            t.record_stream(main)
        return tg, done

    it = iter(groups)
    cur = next(it, None)
    if cur is None:
        return
    cur_t = (compute(cur), None)
    try:
        while cur is not None:
            nxt = next(it, None)
            nxt_t = launch(nxt) if nxt is not None else None
            if cur_t[1] is not None:
                main.wait_event(cur_t[1])
            yield cur, cur_t[0]
            cur, cur_t = nxt, nxt_t
    finally:
        main.wait_stream(part.fwd)            # (also when the consumer abandons the generator: nothing of the second stream outlives the epoch)
