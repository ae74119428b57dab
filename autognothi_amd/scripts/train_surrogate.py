"""Per-batch pieces of reference scripts/train_surrogate.py (:112-211) available on the HIP path:
the uniform mask sampler, the frozen-classifier target forward, the masked surrogate forward and the
KL loss with its gradient w.r.t. the surrogate's output, and the training epoch built on them
(``surrogate_epoch_train``: HIP forward + backward, reference optimiser)."""
from __future__ import annotations

from typing import Any, Callable, Iterable, Optional, Tuple

import torch
from torch import Tensor

from .. import distributed, engine, ops
from ..recipes.types import ModelRecipe
from .common import DROPOUT_RANK_STRIDE, Log, MaskSource, mask_source as common_mask_source, on_epoch_stream, shard


def surrogate_batch_loss(recipe: ModelRecipe, m_classifier, m_surrogate, xs: Tensor, n_players: int, rng,
                         span: Optional[Tuple[int, int, int]] = None):
    """reference :133-149 forward: mask_purely_uniform -> classifier(all ones) -> surrogate(masked) ->
    kl(log_softmax(orig), softmax(adapt)) on the already-softmaxed outputs (the quirk is preserved).
    -> (loss [1], d loss / d adapt_Ys [B,C], orig_Ys, adapt_Ys).  ``xs`` = this rank's inputs, ``span`` = (global inputs,
    lo, hi) their place in the global batch (masks = this rank's rows of the global call); ``rng``: generator or MaskSource."""
    b = xs.shape[0]
    n_tot, lo, hi = span if span is not None else (b, 0, b)
    src = rng if hasattr(rng, "uniform") else MaskSource(rng)
    bits = src.uniform(n_tot, lo, hi, n_players)
    if b == 0:
        return None, None, None, None
    ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
    with torch.no_grad():
        _, orig = recipe.fw_classifier(m_classifier, xs, ones)   # the SECOND output, as the reference (:141): LTT returns (side, backbone)
        adapt, _ = recipe.fw_surrogate(m_surrogate, xs, bits)
    loss, dcur = ops.kl_loss(orig, adapt)
    return loss, dcur, orig, adapt


@on_epoch_stream
def surrogate_epoch_eval(env: Any, device: torch.device, n_players: int, d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe,
                         m_classifier, m_surrogate, epoch: int, gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]],
                         seed: Optional[int] = None, mask_source: Optional[MaskSource] = None) -> float:
    """reference _surrogate_epoch_eval (:163-211) -> mean KL loss.  N > 1 ranks: input slices of every batch by rank, one mask
    stream, the sums reduced once at the end."""
    env = distributed.main_only(env) or Log()
    src = mask_source or common_mask_source(m_surrogate, device, seed)
    _, n_ranks = distributed.world()
    m_classifier.eval(); m_surrogate.eval()
    tot, n = 0.0, 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, _ = gen_input(_inputs, _targets)
        xs, _, n_tot, lo, hi = shard(xs)
        loss, _, _, _ = surrogate_batch_loss(m_recipe, m_classifier, m_surrogate, xs, n_players, src, span=(n_tot, lo, hi))
        if hi == lo:
            continue
        tot += float(loss.item()) * xs.shape[0]
        n += xs.shape[0]
        if n_ranks == 1:
            env.log(f"  > epoch {epoch} :{batch_idx}:test // loss: kl {tot / n:.6f}")
    tot, n = distributed.reduce_scalars([tot, n], device)
    return tot / max(n, 1)


@on_epoch_stream
def surrogate_epoch_train(env: Any, device: torch.device, n_players: int, d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe,
                          m_classifier, m_surrogate, optimizer: torch.optim.Optimizer, epoch: int,
                          gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None,
                          mask_source: Optional[MaskSource] = None) -> float:
    """reference _surrogate_epoch_train (:112-160): uniform masks, frozen-classifier targets (no grad), masked
    surrogate forward + KL + backward on the HIP training kernels, reference optimiser step.  -> mean KL.
    N > 1 ranks: as ``train_explainer.explainer_epoch_train`` — every rank walks the same batches, takes its input slice and its
    rows of the global ``mask_purely_uniform(B, P)`` call, gradients are summed with weights B_r / B (the KL is a batch mean),
    the epoch figure is reduced once."""
    from ..training import make_surrogate_trainer
    env = distributed.main_only(env) or Log()
    src = mask_source or common_mask_source(m_surrogate, device, seed)
    from .. import training as _training
    trainer = m_surrogate.__dict__.get("_ag_trainer") or make_surrogate_trainer(m_recipe, m_surrogate)
    m_surrogate.__dict__["_ag_trainer"] = trainer
    engine.watch_optimizer(optimizer)         # every step() invalidates the weight caches of the parameters it updates
    m_classifier.eval()
    m_surrogate.train()
    tot, n = 0.0, 0
    _, n_ranks = distributed.world()
    reducer = distributed.GradBucketReducer(m_surrogate.parameters()) if n_ranks > 1 else None
    parts = []                                    # device scalars (loss * local batch): read once per epoch
    running = None                                # their running sum (what the per-batch log of the reference prints)
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, _ = gen_input(_inputs, _targets)
        xs, _, n_tot, lo, hi = shard(xs)
        b = hi - lo
        optimizer.zero_grad()
        bits = src.uniform(n_tot, lo, hi, n_players)
        ragged = n_tot < n_ranks
        if reducer is not None:
            reducer.begin(b / float(n_tot))
        _training.GRAD_SINK = reducer.ready if (reducer is not None and not ragged) else None
        try:
            if b > 0:
                ones = torch.ones((b, n_players), dtype=torch.long, device=xs.device)
                with torch.no_grad():
                    _, orig = m_recipe.fw_classifier(m_classifier, xs, ones)   # second output (reference :141)
                # (dropout keys: the same (seed, epoch) on every rank, offset by this rank's first input so that two ranks never
                # draw the same keep pattern for different inputs; lo = 0 at one rank)
                loss, _probs = trainer.loss_and_grads(xs, bits, orig, train=True, seed=(seed or 0) + epoch + DROPOUT_RANK_STRIDE * lo)
                parts.append(loss.reshape(()).float() * b)
                running = parts[-1] if running is None else running + parts[-1]
        finally:
            _training.GRAD_SINK = None
        if reducer is not None:
            reducer.finish(fill_missing=ragged)     # bucketed all-reduces overlapped with the backward (distributed.GradBucketReducer)
        optimizer.step()
        n += b
        if getattr(env, "log_every_step", False) and n_ranks == 1 and b:   # the reference logs every batch (a host read per step)
            env.log(f"  > epoch {epoch} :{batch_idx}:train // loss: kl {float(running.item()) / n:.6f}")
    tot = float(torch.stack(parts).sum().item()) if parts else 0.0
    tot, n = distributed.reduce_scalars([tot, n], device)
    env.log(f"  > epoch {epoch} :train // loss: kl {tot / max(n, 1):.6f}")
    return tot / max(n, 1)


def train_surrogate(env: Any, device: torch.device) -> None:
    """reference train_surrogate(env, device) (scripts/train_surrogate.py:16-109): resume, per epoch reseed -> train epoch ->
    eval epoch -> scheduler step -> metrics -> checkpoint.  ``env`` duck-typed as in scripts/train_explainer.train_explainer."""
    import math
    import time

    from ..utils.tools import set_iterative_seed
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env, save_epoch_ckpt_main
    env = distributed.main_only(env)          # N > 1 ranks: log / metrics / config writes on rank 0 only
    env.log("[[[ train surrogate ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.training.support_surrogate:
        env.log("[[[ skip: surrogate cannot be trained ]]]")
        return
    tcfg = config.train_surrogate
    d_loader = load_cfg_dataset(env, getattr(config, "dataset", None))
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    n_players = m_recipe.n_players(m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    _, m_classifier = load_epoch_model_env(env, m_recipe, "classifier", device=device)
    epoch_surrogate, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate", device=device)
    if epoch_surrogate >= tcfg.epochs:
        env.log("[[[ surrogate already trained ]]]")
        return
    optimizer = torch.optim.AdamW(m_surrogate.parameters(), lr=tcfg.lr, fused=True)   # (same update, single-pass kernel)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, tcfg.epochs)
    for epoch in range(epoch_surrogate + 1, tcfg.epochs + 1):
        seed = set_iterative_seed(config.seed, f"train_surrogate[epoch={epoch}]")
        env.log(f"### epoch {epoch}")
        if getattr(tcfg, "EXPERIMENTAL_progressive_training", None):      # trick for ltt (reference :57-62)
            freeze_lys = min(math.ceil(epoch / 3), m_config.num_hidden_layers)
            env.log(f"  > freeze side branches exc. first {freeze_lys} layers")
            m_surrogate.ltt_freeze_layers_until(freeze_lys)
        ts_begin = time.time()
        train_kld = surrogate_epoch_train(env, device, n_players, d_loader.train(tcfg.batch_size), m_recipe, m_classifier,
                                          m_surrogate, optimizer, epoch, gen_input, seed=seed)
        test_kld = surrogate_epoch_eval(env, device, n_players, d_loader.test(tcfg.batch_size), m_recipe, m_classifier,
                                        m_surrogate, epoch, gen_input)
        scheduler.step()
        ts_delta = time.time() - ts_begin
        if hasattr(env, "metrics"):
            env.metrics({"epoch": epoch, "train_kld_loss": train_kld, "test_kld_loss": test_kld})
        env.log(f"  > epoch {epoch} done in {ts_delta:.2f}s // train_loss: kld {train_kld:.6f} // test_loss: kld {test_kld:.6f}")
        save_epoch_ckpt_main(env.model_path, "surrogate", tcfg, epoch, m_surrogate, env)
