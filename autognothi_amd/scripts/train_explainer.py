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

from .. import distributed, ops
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


def surrogate_targets_lookahead(recipe: ModelRecipe, m_surrogate, xs_list, n_mask_samples: int, n_players: int, rng):
    """The surrogate is frozen while the explainer trains, so the K-mask targets of the NEXT batches do not depend on
    anything the optimiser does: the masked forwards of several consecutive batches are run as ONE forward over their
    concatenated inputs — the hot path then always works on a few thousand rows, whatever the training batch size is (the
    reference trains on 2-4 inputs per step: 64-128 rows, a fraction of one round of GEMM tiles).  Masks are drawn per
    batch, in batch order, from the same device stream, so every batch gets exactly the masks (and values) it would get
    from ``surrogate_targets`` called batch by batch.  -> list of (bits, v_s, v_1) per batch."""
    bits_l = [ops.mask_shapley_new(rng, x.shape[0] * n_mask_samples, n_players, want_i64=False, want_bits=True)[1] for x in xs_list]
    xs_all = torch.cat(list(xs_list), dim=0) if len(xs_list) > 1 else xs_list[0]
    bits_all = torch.cat(bits_l, dim=0) if len(bits_l) > 1 else bits_l[0]
    m_surrogate.eval()
    with torch.no_grad():
        v_s, _ = recipe.fw_surrogate(m_surrogate, xs_all, bits_all)
        ones = torch.ones((xs_all.shape[0], n_players), dtype=torch.long, device=xs_all.device)
        v_1, _ = recipe.fw_surrogate(m_surrogate, xs_all, ones)
    out, r0, b0 = [], 0, 0
    for x, bits in zip(xs_list, bits_l):
        b = x.shape[0]
        out.append((bits, v_s[r0:r0 + b * n_mask_samples], v_1[b0:b0 + b]))
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


def explainer_epoch_train(env: Any, device: torch.device, n_mask_samples: int, n_players: int, v_0: Tensor,
                          d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, m_explainer,
                          optimizer: torch.optim.Optimizer, epoch: int,
                          gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]], seed: Optional[int] = None,
                          target_rows: int = 1536) -> float:
    """reference _explainer_epoch_train (:128-207) / _duo_explainer_epoch_train: per batch — K-mask surrogate
    targets (no grad, HIP inference path), explainer forward + Shapley loss + backward (HIP training kernels,
    autognothi_amd/training.py), then the reference's own optimiser step.  -> train_reg_loss (mean)."""
    from ..training import make_explainer_trainer
    env = env or Log()
    rng = device_rng(m_surrogate, device, seed)
    from .. import training as _training
    trainer = m_explainer.__dict__.get("_ag_trainer") or make_explainer_trainer(m_recipe, m_explainer)
    m_explainer.__dict__["_ag_trainer"] = trainer
    total = 0
    losses = []                                   # device scalars: read back ONCE per epoch (no per-step host sync)
    m_explainer.train()
    # N > 1 ranks (rows sharded by input): gradients are averaged over RCCL in 64 MiB buckets whose all-reduce starts as soon
    # as the backward has finished them (distributed.GradBucketReducer); a no-op at N = 1
    _, n_ranks = distributed.world()
    reducer = distributed.GradBucketReducer(m_explainer.parameters()) if n_ranks > 1 else None
    # surrogate targets are computed for groups of consecutive batches at once (surrogate_targets_lookahead): as many batches
    # as it takes to reach `target_rows` masked rows per forward (1536 = 48 inputs x 32 masks, the size the kernels are
    # tuned for); target_rows = 0 computes them batch by batch
    def grouped(items):
        group, rows = [], 0
        for idx, (_inputs, _targets) in enumerate(items):
            xs_, zs_ = gen_input(_inputs, _targets)
            group.append((idx, xs_, zs_))
            rows += xs_.shape[0] * n_mask_samples
            if rows >= target_rows:
                yield group
                group, rows = [], 0
        if group:
            yield group

    def batches():
        for group in grouped(d_items):
            tg = surrogate_targets_lookahead(m_recipe, m_surrogate, [g_[1] for g_ in group], n_mask_samples, n_players, rng)
            for (idx, xs_, zs_), t_ in zip(group, tg):
                yield idx, xs_, zs_, t_

    for batch_idx, xs, zs, (bits, v_s, v_1) in batches():
        optimizer.zero_grad()
        _training.GRAD_SINK = reducer.ready if reducer is not None else None
        try:
            loss, _phi = trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, n_mask_samples, labels=zs, train=True,
                                                seed=(seed or 0) + epoch)
        finally:
            _training.GRAD_SINK = None
        if reducer is not None:
            reducer.finish()
        optimizer.step()
        losses.append(loss.reshape(()))
        total += xs.shape[0]
        if getattr(env, "log_every_step", False):  # the reference logs the loss of every batch (a host read per step)
            env.log(f"  > epoch {epoch} :{batch_idx}:train // loss: shap {float(loss.item()) / xs.shape[0]:.6f}, fin {total}")
    reg_loss = float(torch.stack(losses).sum().item()) if losses else 0.0
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
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env, save_epoch_ckpt_cfg
    env.log("[[[ train explainer ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.training.support_explainer:
        env.log("[[[ skip: explainer cannot be trained ]]]")
        return
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
        if save_epoch_ckpt_cfg(env.model_path, "explainer", tcfg, epoch, m_explainer) and hasattr(env, "flush_cfg"):
            env.flush_cfg()
