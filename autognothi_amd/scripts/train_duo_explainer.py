"""Classifier + explainer ("duo") training on the HIP path (reference scripts/train_duo_explainer.py): the same batch loop as
scripts/train_explainer.py with the classification head trained beside the Shapley head — loss = cross_entropy(base_Ys, Zs)
+ loss_shapley_new(...) (:180-196; duo-ViT feeds probabilities to the cross entropy, duo-BERT raw logits) — and
the reference's four per-epoch figures (classification loss, Shapley loss, their sum, classification accuracy)."""
from __future__ import annotations

import time
from typing import Any, Callable, Iterable, List, Optional, Tuple

import torch
from torch import Tensor

from .. import distributed, engine, ops
from ..recipes.types import ModelRecipe
from .common import DROPOUT_RANK_STRIDE, Log, MaskSource, mask_source as common_mask_source, on_epoch_stream, shard_auto, pipelined_targets, train_partition_all_ranks, log_schedule
from .train_explainer import explainer_batch_loss, surrogate_null, surrogate_targets, surrogate_targets_lookahead


def _cross_entropy_value(base: Tensor, labels: Tensor) -> Tensor:
    """F.cross_entropy(base_Ys, Zs) (mean; :178, :270) from the soft-max kernel."""
    s = ops.softmax_rows(base.float().contiguous())
    return -(s.gather(1, labels.view(-1, 1).to(torch.int64)).log()).mean()


@on_epoch_stream
def duo_explainer_epoch_train(env: Any, device: torch.device, n_mask_samples: int, n_players: int, v_0: Tensor,
                              d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, m_explainer,
                              optimizer: torch.optim.Optimizer, epoch: int,
                              gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None,
                              target_rows: int = int(__import__('os').environ.get('AG_TARGET_ROWS', '1536')), mask_source: Optional[MaskSource] = None) -> Tuple[float, float, float, float]:
    """reference _duo_explainer_epoch_train (:121-213) -> (train_cls_loss, train_reg_loss, train_loss, train_cls_acc), the
    three losses as the reference accumulates them (sum of the per-batch values / samples).  The losses and the hit count stay
    on the device during the epoch and are read once at its end.  N > 1 ranks: sharded exactly as
    ``train_explainer.explainer_epoch_train`` (inputs of every batch by rank, one mask stream, gradients summed with weights
    B_r / B, the epoch figures reduced once)."""
    from .. import training as _training
    from ..training import make_explainer_trainer
    env = distributed.main_only(env) or Log()
    src = mask_source or common_mask_source(m_surrogate, device, seed)
    trainer = m_explainer.__dict__.get("_ag_trainer") or make_explainer_trainer(m_recipe, m_explainer)
    m_explainer.__dict__["_ag_trainer"] = trainer
    if not getattr(trainer, "duo", False):
        raise ValueError("duo_explainer_epoch_train: the explainer has no classification head (not a duo recipe)")
    engine.watch_optimizer(optimizer)         # every step() invalidates the weight caches of the parameters it updates
    m_explainer.train()
    _, n_ranks = distributed.world()
    reducer = distributed.GradBucketReducer(m_explainer.parameters()) if n_ranks > 1 else None
    parts: List[Tensor] = []          # [cls, shap, correct] per batch
    total = 0

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

    # (one rank: the targets of the NEXT group on a second stream beside this group's steps, scripts/common.TrainPartition)
    part = train_partition_all_ranks(device, m_explainer)
    log_schedule(env, part)

    def compute(group):
        return surrogate_targets_lookahead(m_recipe, m_surrogate, [g_[1] for g_ in group], n_mask_samples, n_players, src,
                                           spans=[g_[3] for g_ in group])

    for group, tg in pipelined_targets(grouped(d_items), compute, part):
        for (batch_idx, xs, zs, sp), (bits, v_s, v_1) in zip(group, tg):
            optimizer.zero_grad()
            n_tot, lo, hi = sp.astuple()
            if sp.by_mask:
                # fewer inputs than ranks (common.shard_auto): targets sharded by mask and gathered; every rank takes the same
                # step on the whole batch — no gradient exchange, no idle rank; each rank reports 1 / ranks of the batch
                trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, n_mask_samples, labels=zs, train=True, seed=(seed or 0) + epoch)
                optimizer.step()
                l_shap, l_cls, base = trainer.last_parts
                hits = base.argmax(dim=1).eq(zs.to(base.device)).sum().float()
                parts.append(torch.stack([l_cls.reshape(()).float(), l_shap.reshape(()).float(), hits]) / n_ranks)
                total += n_tot / float(n_ranks)
                continue
            weight = (hi - lo) / float(n_tot)
            ragged = n_tot < n_ranks
            if reducer is not None:
                reducer.begin(weight)
            _training.GRAD_SINK = reducer.ready if (reducer is not None and not ragged) else None
            try:
                if hi > lo:
                    trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, n_mask_samples, labels=zs, train=True,
                                           seed=(seed or 0) + epoch + DROPOUT_RANK_STRIDE * lo)
            finally:
                _training.GRAD_SINK = None
            if reducer is not None:
                reducer.finish(fill_missing=ragged)
            optimizer.step()
            if hi == lo:
                continue
            l_shap, l_cls, base = trainer.last_parts
            hits = base.argmax(dim=1).eq(zs.to(base.device)).sum().float()
            wl = 1.0 if n_ranks == 1 else weight      # the two losses are batch means: this rank's share of the global mean
            parts.append(torch.stack([l_cls.reshape(()).float() * wl, l_shap.reshape(()).float() * wl, hits]))
            total += xs.shape[0]
            if getattr(env, "log_every_step", False) and n_ranks == 1:   # the reference logs every batch (three host reads per step)
                c_, s_, h_ = [float(v) for v in parts[-1].tolist()]
                env.log(f"  > epoch {epoch} :{batch_idx}:train // loss: cls {c_ / xs.shape[0]:.6f} shap {s_ / xs.shape[0]:.6f} "
                        f"tot {(c_ + s_) / xs.shape[0]:.6f}")
    sums = [float(v) for v in torch.stack(parts).sum(0).tolist()] if parts else [0.0, 0.0, 0.0]
    cls_loss, reg_loss, correct, total = distributed.reduce_scalars(sums + [total], device)
    total = int(round(total))
    if total == 0:
        return 0.0, 0.0, 0.0, 0.0
    env.log(f"  > epoch {epoch} :train // loss: cls {cls_loss / total:.6f} shap {reg_loss / total:.6f} "
            f"tot {(cls_loss + reg_loss) / total:.6f} // acc: {100.0 * correct / total:.3f}%, {int(correct)}/{total}")
    return cls_loss / total, reg_loss / total, (cls_loss + reg_loss) / total, correct / total


@on_epoch_stream
def duo_explainer_epoch_eval(env: Any, device: torch.device, n_mask_samples: int, n_players: int, v_0: Tensor,
                             d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, m_explainer, epoch: int,
                             gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None,
                             mask_source: Optional[MaskSource] = None) -> Tuple[float, float, float, float, List[Any]]:
    """reference _duo_explainer_epoch_eval (:216-307) -> (test_cls_loss, test_reg_loss, test_loss, test_cls_acc, test_plots);
    the plots list is empty there too (":todo: make plots")."""
    env = distributed.main_only(env) or Log()
    src = mask_source or common_mask_source(m_surrogate, device, seed)
    _, n_ranks = distributed.world()
    m_explainer.eval()
    parts: List[Tensor] = []
    total = 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, zs = gen_input(_inputs, _targets)
        xs, zs, sp = shard_auto(xs, zs, n_mask_samples)
        n_tot, lo, hi = sp.astuple()
        bits, v_s, v_1 = surrogate_targets(m_recipe, m_surrogate, xs, n_mask_samples, n_players, src, span=sp)
        if hi == lo:
            continue
        wl = 1.0 if n_ranks == 1 else ((1.0 / n_ranks) if sp.by_mask else (hi - lo) / float(n_tot))
        l_shap, _, _, base = explainer_batch_loss(m_recipe, m_explainer, xs, bits, v_0, v_s, v_1, n_mask_samples, n_players)
        if base is None:
            raise ValueError("duo_explainer_epoch_eval: fw_explainer returned no class output (not a duo recipe)")
        l_cls = _cross_entropy_value(base, zs.to(base.device))
        hits = base.argmax(dim=1).eq(zs.to(base.device)).sum().float()
        if sp.by_mask:      # every rank evaluated the whole batch: each reports 1 / ranks of its hits and samples
            hits = hits / n_ranks
        parts.append(torch.stack([l_cls.reshape(()).float() * wl, l_shap.reshape(()).float() * wl, hits]))
        total += (n_tot / float(n_ranks)) if sp.by_mask else xs.shape[0]
    sums = [float(v) for v in torch.stack(parts).sum(0).tolist()] if parts else [0.0, 0.0, 0.0]
    cls_loss, reg_loss, correct, total = distributed.reduce_scalars(sums + [total], device)
    total = int(round(total))
    if total == 0:
        return 0.0, 0.0, 0.0, 0.0, []
    env.log(f"  > epoch {epoch} :test // loss: cls {cls_loss / total:.6f} shap {reg_loss / total:.6f} "
            f"tot {(cls_loss + reg_loss) / total:.6f} // acc: {100.0 * correct / total:.3f}%, {int(correct)}/{total}")
    return cls_loss / total, reg_loss / total, (cls_loss + reg_loss) / total, correct / total, []


def train_duo_explainer(env: Any, device: torch.device) -> None:
    """reference train_duo_explainer(env, device) (:20-118): resume, per-epoch reseed ("train_explainer[epoch=E]": the duo
    script shares the key, :56), train + eval epoch, cosine schedule, the ten-field metrics entry, checkpoint.  ``env`` is
    duck-typed as in scripts/train_explainer.train_explainer."""
    from ..utils.tools import set_iterative_seed
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env, save_epoch_ckpt_main
    env = distributed.main_only(env)
    env.log("[[[ !!! *experimental* train (duo) classifier + explainer !!! ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.training.support_explainer or not m_recipe.training.exp_variant_duo:
        env.log("[[[ skip: explainer cannot be trained ]]]")
        return
    tcfg = config.train_explainer
    d_loader = load_cfg_dataset(env, getattr(config, "dataset", None))
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    n_players = m_recipe.n_players(m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    _, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate", device=device)
    epoch_explainer, m_explainer = load_epoch_model_env(env, m_recipe, "explainer", device=device)
    optimizer = torch.optim.AdamW(m_explainer.parameters(), lr=tcfg.lr, fused=True)    # (:39-41; single-pass fused form)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, tcfg.epochs)
    v_0 = surrogate_null(m_recipe, m_config, m_misc, m_surrogate, device)
    for epoch in range(epoch_explainer + 1, tcfg.epochs + 1):
        seed = set_iterative_seed(config.seed, f"train_explainer[epoch={epoch}]")
        env.log(f"### epoch {epoch}")
        ts_begin = time.time()
        tr = duo_explainer_epoch_train(env, device, tcfg.n_mask_samples, n_players, v_0, d_loader.train(tcfg.batch_size),
                                       m_recipe, m_surrogate, m_explainer, optimizer, epoch, gen_input, seed=seed)
        te = duo_explainer_epoch_eval(env, device, tcfg.n_mask_samples, n_players, v_0, d_loader.test(tcfg.batch_size),
                                      m_recipe, m_surrogate, m_explainer, epoch, gen_input)
        scheduler.step()
        ts_delta = time.time() - ts_begin
        if hasattr(env, "metrics"):
            env.metrics({"epoch": epoch, "train_cls_loss": tr[0], "train_reg_loss": tr[1], "train_loss": tr[2], "train_cls_acc": tr[3],
                         "test_cls_loss": te[0], "test_reg_loss": te[1], "test_loss": te[2], "test_cls_acc": te[3],
                         "test_plots": te[4]})
        env.log(f"  > epoch {epoch} done in {ts_delta:.2f}s // train_loss: shap {tr[1]:.6f} // test_loss: shap {te[1]:.6f}")
        save_epoch_ckpt_main(env.model_path, "explainer", tcfg, epoch, m_explainer, env)
