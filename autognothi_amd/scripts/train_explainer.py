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

from .. import distributed, engine, ops
from ..recipes.types import ModelRecipe
from .common import DROPOUT_RANK_STRIDE, Log, MaskSource, Span, device_rng, on_epoch_stream, pipelined_targets, shard, shard_auto, train_partition_all_ranks, log_schedule
from .common import mask_source as common_mask_source


def surrogate_null(recipe: ModelRecipe, cfg, misc, m_surrogate, device: torch.device) -> Tensor:
    """reference scripts/train_explainer.py:55-60."""
    n_players = recipe.n_players(cfg)
    null_xs = recipe.gen_null(cfg, misc, device)
    m_surrogate.eval()
    with torch.no_grad():
        v0, _ = recipe.fw_surrogate(m_surrogate, null_xs, torch.ones((1, n_players), dtype=torch.long, device=device))
    return v0


def _source(rng_or_source) -> MaskSource:
    return rng_or_source if hasattr(rng_or_source, "shapley") else MaskSource(rng_or_source)


def _as_span(span, b: int, k: int) -> Span:
    if isinstance(span, Span):
        return span
    n_total, lo, hi = span if span is not None else (b, 0, b)
    return Span(n_total, lo, hi, "input", k)


def _targets_by_mask(recipe: ModelRecipe, m_surrogate, xs: Tensor, k: int, n_players: int, src: MaskSource, sp: Span, bits=None):
    """Fewer inputs than ranks (the per-image loop of scripts/measure_faithfulness.py:195-218 and BASELINE config 4 at
    one input per step): every rank holds ALL ``sp.n_tot`` inputs and runs masks [k_lo, k_hi) of each — its share of the hot
    path — then the rows are all-gathered back into the reference's input-major order (``gather_masks_within_inputs``: a few
    KB).  -> (mask bits of the WHOLE batch, v_s of the whole batch, v_1): every rank continues with the complete targets, so the
    explainer step that follows is the same on all ranks and needs no gradient exchange."""
    bits = src.shapley(sp.n_tot, 0, sp.n_tot, k, n_players) if bits is None else bits    # the whole global call: the one mask stream
    width = bits.shape[-1]
    local = bits.view(sp.n_tot, k, width)[:, sp.k_lo:sp.k_hi].reshape(-1, width).contiguous()
    m_surrogate.eval()
    with torch.no_grad():
        v_loc, _ = recipe.fw_surrogate(m_surrogate, xs, local)       # n_tot inputs x (k_hi - k_lo) masks: shared layer 0
        ones = torch.ones((xs.shape[0], n_players), dtype=torch.long, device=xs.device)
        v_1, _ = recipe.fw_surrogate(m_surrogate, xs, ones)
    v_s = distributed.gather_masks_within_inputs(v_loc.contiguous(), sp.n_tot, k)
    return bits, v_s, v_1


def surrogate_targets(recipe: ModelRecipe, m_surrogate, xs: Tensor, n_mask_samples: int, n_players: int, rng,
                      span=None) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """-> (mask key bits [B*K, Tw], v_s [B*K, C], v_1 [B, C]); row order [b0 s0, b0 s1, b1 s0, ...].  ``xs`` are THIS rank's
    inputs; ``span`` = (inputs of the global batch, lo, hi) or a ``common.Span`` names them inside the global batch (default: xs
    is the batch) — the masks are this rank's rows of the global ``mask_shapley_new`` call (scripts/common.MaskSource).  A span
    in mask mode (``common.shard_auto``: fewer inputs than ranks) shards the K masks of every input instead and returns the
    gathered targets of the whole batch.  ``rng``: a device generator or a MaskSource.  An empty shard returns
    (bits [0, Tw], None, None)."""
    b = xs.shape[0]
    sp = _as_span(span, b, n_mask_samples)
    if sp.by_mask:
        return _targets_by_mask(recipe, m_surrogate, xs, n_mask_samples, n_players, _source(rng), sp)
    bits = _source(rng).shapley(sp.n_tot, sp.lo, sp.hi, n_mask_samples, n_players)
    if b == 0:
        return bits, None, None
    m_surrogate.eval()
    with torch.no_grad():
        v_s, _ = recipe.fw_surrogate(m_surrogate, xs, bits)          # B inputs, B*K mask rows: shared layer 0
        ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
        v_1, _ = recipe.fw_surrogate(m_surrogate, xs, ones)
    return bits, v_s, v_1


def surrogate_targets_lookahead(recipe: ModelRecipe, m_surrogate, xs_list, n_mask_samples: int, n_players: int, rng, spans=None):
    """The surrogate is frozen while the explainer trains, so the K-mask targets of the NEXT batches do not depend on
    anything the optimiser does: the masked forwards of several consecutive batches are run as ONE forward over their
    concatenated inputs — the hot path then always works on a few thousand rows, whatever the training batch size is (the
    reference trains on 2-4 inputs per step: 64-128 rows, a fraction of one round of GEMM tiles).  Masks are drawn per
    batch, in batch order, from the same stream, so every batch gets exactly the masks (and values) it would get
    from ``surrogate_targets`` called batch by batch.  ``spans[i]`` = (global inputs, lo, hi) or a ``common.Span`` of batch i for
    row-sharded ranks (xs_list holds the local slices); batches in mask mode (fewer inputs than ranks) run on their own —
    their rows per input differ from the others' — and come back with the gathered targets of the whole batch.
    -> list of (bits, v_s, v_1) per batch."""
    src = _source(rng)
    spans = spans if spans is not None else [(x.shape[0], 0, x.shape[0]) for x in xs_list]
    spans = [_as_span(sp, x.shape[0], n_mask_samples) for sp, x in zip(spans, xs_list)]
    # every batch's masks first, in batch order: ONE stream (a mask-mode batch takes the whole global call)
    bits_l = [src.shapley(sp.n_tot, 0 if sp.by_mask else sp.lo, sp.n_tot if sp.by_mask else sp.hi, n_mask_samples, n_players) for sp in spans]
    out = [(bits_l[i], None, None) for i in range(len(xs_list))]
    for i, sp in enumerate(spans):
        if sp.by_mask:
            out[i] = _targets_by_mask(recipe, m_surrogate, xs_list[i], n_mask_samples, n_players, src, sp, bits=bits_l[i])
    live = [i for i, x in enumerate(xs_list) if x.shape[0] > 0 and not spans[i].by_mask]
    if not live:
        return out
    xs_all = torch.cat([xs_list[i] for i in live], dim=0) if len(live) > 1 else xs_list[live[0]]
    bits_all = torch.cat([bits_l[i] for i in live], dim=0) if len(live) > 1 else bits_l[live[0]]
    m_surrogate.eval()
    with torch.no_grad():
        v_s, _ = recipe.fw_surrogate(m_surrogate, xs_all, bits_all)
        ones = torch.ones((xs_all.shape[0], n_players), dtype=torch.long, device=xs_all.device)
        v_1, _ = recipe.fw_surrogate(m_surrogate, xs_all, ones)
    r0, b0 = 0, 0
    for i in live:
        b = xs_list[i].shape[0]
        out[i] = (bits_l[i], v_s[r0:r0 + b * n_mask_samples], v_1[b0:b0 + b])
        r0 += b * n_mask_samples
        b0 += b
    return out


def explainer_batch_loss(recipe: ModelRecipe, m_explainer, xs: Tensor, bits: Tensor, v_0: Tensor, v_s: Tensor, v_1: Tensor,
                         n_mask_samples: int, n_players: int, want_grad: bool = False):
    """reference :184-196 (forward): -> (loss [1] device tensor, phi [B,C,P], dphi or None, logits or None)."""
    b = xs.shape[0]
    ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
    with torch.no_grad():
        phi, logits = recipe.fw_explainer(m_explainer, xs, ones, v_1, v_0)
    loss, dphi = ops.shapley_loss(bits, v_0, v_s, phi, b, n_mask_samples, want_grad=want_grad)
    return loss, phi, dphi, logits


@on_epoch_stream
def explainer_epoch_eval(env: Any, device: torch.device, n_mask_samples: int, n_players: int, v_0: Tensor,
                         d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, m_explainer, epoch: int,
                         gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None,
                         mask_source: Optional[MaskSource] = None) -> float:
    """reference _explainer_epoch_eval (:210-281) -> test_reg_loss (sum of the per-batch mean losses / samples, as the
    reference accumulates it).  N > 1 ranks: every rank walks the same batches and takes its input slice (as in
    ``explainer_epoch_train``); the epoch figure is reduced once at the end."""
    env = distributed.main_only(env) or Log()
    src = mask_source or common_mask_source(m_surrogate, device, seed)
    reg_loss, total = 0.0, 0
    m_explainer.eval()
    _, n_ranks = distributed.world()
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, _zs = gen_input(_inputs, _targets)
        xs, _zs, sp = shard_auto(xs, _zs, n_mask_samples)
        n_tot, lo, hi = sp.astuple()
        bits, v_s, v_1 = surrogate_targets(m_recipe, m_surrogate, xs, n_mask_samples, n_players, src, span=sp)
        if hi == lo:
            continue
        loss, _, _, _ = explainer_batch_loss(m_recipe, m_explainer, xs, bits, v_0, v_s, v_1, n_mask_samples, n_players)
        # this rank's share of the global batch-mean loss (mask mode: every rank computed the whole batch's loss)
        share = (1.0 / n_ranks) if sp.by_mask else ((hi - lo) / n_tot)
        lv = float(loss.item()) * share
        reg_loss += lv
        total += (n_tot / n_ranks) if sp.by_mask else (hi - lo)
        if distributed.world()[1] == 1:
            env.log(f"  > epoch {epoch} :{batch_idx}:test // loss: shap {lv / xs.shape[0]:.6f}, fin {total}")
    reg_loss, total = distributed.reduce_scalars([reg_loss, total], device)
    total = int(round(total))
    return reg_loss / max(total, 1)


@on_epoch_stream
def explainer_epoch_train(env: Any, device: torch.device, n_mask_samples: int, n_players: int, v_0: Tensor,
                          d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, m_explainer,
                          optimizer: torch.optim.Optimizer, epoch: int,
                          gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None,
                          target_rows: int = int(__import__('os').environ.get('AG_TARGET_ROWS', '1536')), mask_source: Optional[MaskSource] = None) -> float:
    """reference _explainer_epoch_train (:128-207) / _duo_explainer_epoch_train: per batch — K-mask surrogate
    targets (no grad, HIP inference path), explainer forward + Shapley loss + backward (HIP training kernels,
    autognothi_amd/training.py), then the reference's own optimiser step.  -> train_reg_loss (mean).

    N > 1 ranks (one process per GPU; BASELINE config 5).  ``d_items`` yields the SAME global batches on every
    rank (the reference's loader, unchanged); rank r takes inputs ``distributed.shard_range(B)`` of every batch together
    with all K of their masks — its rows of the one global ``mask_shapley_new(B*K, P)`` call (``MaskSource``: no traffic,
    bit-identical to the single-process stream) — runs targets, forward and backward on them, and the gradients are summed
    over the ranks weighted by B_r / B (``GradBucketReducer``: 64 MiB buckets, each all-reduce in flight while the backward
    below it still runs), which is exactly the gradient of the reference's batch-mean loss over the B inputs.  The epoch loss
    is reduced ONCE at the end.  A batch with fewer inputs than ranks (the ragged tail of an epoch; a one-input step of BASELINE
    config 4) is sharded by MASK instead (``common.shard_auto``): every rank runs masks [k_lo, k_hi) of every input through the
    surrogate, the targets are all-gathered (a few KB) and every rank takes the same explainer step on the whole batch — no
    rank idles and no gradient travels.  (Only when K < ranks do ranks go without work: they then enter the collectives with
    zero gradients, in parameter order.)  With one rank this is the reference loop, step for step."""
    from ..training import make_explainer_trainer
    env = distributed.main_only(env) or Log()
    src = mask_source or common_mask_source(m_surrogate, device, seed)
    from .. import training as _training
    trainer = m_explainer.__dict__.get("_ag_trainer") or make_explainer_trainer(m_recipe, m_explainer)
    m_explainer.__dict__["_ag_trainer"] = trainer
    engine.watch_optimizer(optimizer)         # every step() invalidates the weight caches of the parameters it updates
    total = 0
    losses = []                                   # device scalars: read back ONCE per epoch (no per-step host sync)
    m_explainer.train()
    _, n_ranks = distributed.world()
    reducer = distributed.GradBucketReducer(m_explainer.parameters()) if n_ranks > 1 else None
    # surrogate targets are computed for groups of consecutive batches at once (surrogate_targets_lookahead): as many batches
    # as it takes to reach `target_rows` masked rows per forward (1536 = 48 inputs x 32 masks, the size the kernels are
    # tuned for); target_rows = 0 computes them batch by batch

    def grouped(items):
        group, rows = [], 0
        for idx, (_inputs, _targets) in enumerate(items):
            xs_, zs_ = gen_input(_inputs, _targets)
            xs_, zs_, sp_ = shard_auto(xs_, zs_, n_mask_samples)
            group.append((idx, xs_, zs_, sp_))
            rows += xs_.shape[0] * ((sp_.k_hi - sp_.k_lo) if sp_.by_mask else n_mask_samples)
            if rows >= target_rows:
                yield group
                group, rows = [], 0
        if group:
            yield group

    # on a GPU the target forward of the NEXT group runs on a second stream (its persistent GEMM on 3/4 of every XCD's CUs) while this
    # group's steps run on the caller's (common.TrainPartition / pipelined_targets) — with N > 1 ranks beside the gradient exchange too;
    # otherwise (AG_TRAIN_PARTITION=0, no second hardware queue) the two alternate
    part = train_partition_all_ranks(device, m_explainer)
    log_schedule(env, part)

    def batches():
        def compute(group):
            return surrogate_targets_lookahead(m_recipe, m_surrogate, [g_[1] for g_ in group], n_mask_samples, n_players, src,
                                               spans=[g_[3] for g_ in group])
        for group, tg in pipelined_targets(grouped(d_items), compute, part):
            for (idx, xs_, zs_, span), t_ in zip(group, tg):
                yield idx, xs_, zs_, span, t_

    def one_step(batch_idx, xs, zs, sp, bits, v_s, v_1):
        nonlocal total
        optimizer.zero_grad()
        n_tot, lo, hi = sp.astuple()
        if sp.by_mask:
            # fewer inputs than ranks: the K-mask targets were sharded by mask and gathered, every rank holds the whole batch —
            # the same explainer step on all ranks (same inputs, targets, dropout keys), so no gradient exchange and no idle rank
            loss, _phi = trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, n_mask_samples, labels=zs, train=True, seed=(seed or 0) + epoch)
            optimizer.step()
            losses.append(loss.reshape(()) / n_ranks)
            total += n_tot / float(n_ranks)
            return
        weight = (hi - lo) / float(n_tot)
        ragged = n_tot < n_ranks                  # some rank holds no input of this batch: un-instrumented exchange for all
        if reducer is not None:
            reducer.begin(weight)
        _training.GRAD_SINK = reducer.ready if (reducer is not None and not ragged) else None
        try:
            if hi > lo:
                loss, _phi = trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, n_mask_samples, labels=zs, train=True,
                                                    seed=(seed or 0) + epoch + DROPOUT_RANK_STRIDE * lo)
                loss = loss.reshape(())
            else:
                loss = torch.zeros((), dtype=torch.float32, device=v_0.device)
        finally:
            _training.GRAD_SINK = None
        if reducer is not None:
            reducer.finish(fill_missing=ragged)
        optimizer.step()
        losses.append(loss if n_ranks == 1 else loss * weight)
        total += hi - lo
        if getattr(env, "log_every_step", False) and n_ranks == 1:  # the reference logs the loss of every batch (a host read per step)
            env.log(f"  > epoch {epoch} :{batch_idx}:train // loss: shap {float(loss.item()) / xs.shape[0]:.6f}, fin {total}")

    for batch_idx, xs, zs, sp, (bits, v_s, v_1) in batches():
        one_step(batch_idx, xs, zs, sp, bits, v_s, v_1)
    reg_loss = float(torch.stack(losses).sum().item()) if losses else 0.0
    reg_loss, total = distributed.reduce_scalars([reg_loss, total], device)
    total = int(round(total))
    env.log(f"  > epoch {epoch} :train // loss: shap {reg_loss / max(total, 1):.6f}, fin {total}")
    return reg_loss / max(total, 1)


def train_explainer(env: Any, device: torch.device) -> None:
    """reference train_explainer(env, device) (scripts/train_explainer.py:19-125; duo recipes: train_duo_explainer.py:20-118):
    resume from the newest explainer checkpoint, per epoch reseed (set_iterative_seed) -> train epoch -> eval epoch ->
    scheduler step -> metrics -> checkpoint.  ``env`` is duck-typed: ``.config`` (``net``, ``seed``, ``train_explainer``
    with epochs / lr / batch_size / n_mask_samples / ckpt_when), ``.model_path``, ``.log``, and optionally ``.metrics``,
    ``.flush_cfg``, ``.d_loader`` (scripts/resources.load_cfg_dataset).  The kernel_shap variant is out of scope."""
    import math
    import time

    from ..utils.tools import set_iterative_seed
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env, save_epoch_ckpt_main
    env = distributed.main_only(env)          # N > 1 ranks: log / metrics / config writes on rank 0 only
    env.log("[[[ train explainer ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.training.support_explainer:
        env.log("[[[ skip: explainer cannot be trained ]]]")
        return
    if m_recipe.training.exp_variant_duo:     # reference :26-27: duo recipes have their own loop and metrics
        from .train_duo_explainer import train_duo_explainer
        return train_duo_explainer(env, device)
    if m_recipe.training.exp_variant_kernel_shap:
        raise NotImplementedError("the kernel_shap explainer baseline is outside this build's scope")
    tcfg = config.train_explainer
    d_loader = load_cfg_dataset(env, getattr(config, "dataset", None))
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    n_players = m_recipe.n_players(m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    _, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate", device=device)
    epoch_explainer, m_explainer = load_epoch_model_env(env, m_recipe, "explainer", device=device)
    if epoch_explainer >= tcfg.epochs:
        env.log("[[[ explainer already trained ]]]")
        return
    # the reference's optimiser (scripts/train_explainer.py:47-49: AdamW, default betas / eps / weight decay) in torch's
    # single-pass fused form: the same update, one kernel per parameter group instead of eight elementwise passes
    optimizer = torch.optim.AdamW(m_explainer.parameters(), lr=tcfg.lr, fused=True)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, tcfg.epochs)
    v_0 = surrogate_null(m_recipe, m_config, m_misc, m_surrogate, device)
    for epoch in range(epoch_explainer + 1, tcfg.epochs + 1):
        seed = set_iterative_seed(config.seed, f"train_explainer[epoch={epoch}]")
        env.log(f"### epoch {epoch}")
        if getattr(tcfg, "EXPERIMENTAL_progressive_training", None):      # trick for ltt (reference :69-74)
            freeze_lys = min(math.ceil(epoch / 2), m_config.num_hidden_layers)
            env.log(f"  > freeze side branches exc. first {freeze_lys} layers")
            m_explainer.ltt_freeze_layers_until(freeze_lys)
        ts_begin = time.time()
        train_reg_loss = explainer_epoch_train(env, device, tcfg.n_mask_samples, n_players, v_0, d_loader.train(tcfg.batch_size),
                                               m_recipe, m_surrogate, m_explainer, optimizer, epoch, gen_input, seed=seed)
        test_reg_loss = explainer_epoch_eval(env, device, tcfg.n_mask_samples, n_players, v_0, d_loader.test(tcfg.batch_size),
                                             m_recipe, m_surrogate, m_explainer, epoch, gen_input)
        scheduler.step()
        ts_delta = time.time() - ts_begin
        if hasattr(env, "metrics"):
            env.metrics({"epoch": epoch, "train_reg_loss": train_reg_loss, "test_reg_loss": test_reg_loss, "test_plots": []})
        env.log(f"  > epoch {epoch} done in {ts_delta:.2f}s // train_loss: shap {train_reg_loss:.6f} // "
                f"test_loss: shap {test_reg_loss:.6f}")
        save_epoch_ckpt_main(env.model_path, "explainer", tcfg, epoch, m_explainer, env)
