"""Classifier training on the HIP path (reference scripts/train_classifier.py): per batch fw_classifier on the all-ones mask,
cross entropy on the class output, backward, optimiser step (:117-147); per epoch reseed, eval, cosine schedule, metrics,
checkpoint (:15-104).  The vanilla classifiers freeze themselves in ``train()`` (models/vanilla_vit.py:46-50): as in the
reference an epoch only trains what ``set_model_mode`` (scripts/pretrain_classifier.py:27-48) or the recipe leaves trainable."""
from __future__ import annotations

import math
import time
from typing import Any, Callable, Iterable, Optional, Tuple

import torch
from torch import Tensor

from .. import distributed, engine
from ..recipes.types import ModelRecipe
from .common import Log


def _ce(base: Tensor, labels: Tensor) -> Tensor:
    from .train_duo_explainer import _cross_entropy_value
    return _cross_entropy_value(base, labels)


def classifier_epoch_train(env: Any, device: torch.device, n_players: int, d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe,
                           m_classifier, m_classifier_set_mode: Optional[Callable[[Any, bool], None]],
                           optimizer: torch.optim.Optimizer, epoch: int, gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]],
                           seed: Optional[int] = None) -> Tuple[float, float]:
    """reference _classifier_epoch_train (:107-150) -> (train_cls_loss, train_cls_acc)."""
    from .. import training as _training
    from ..training import _cross_entropy, make_surrogate_trainer
    env = env or Log()
    m_classifier.train()
    if m_classifier_set_mode is not None:
        m_classifier_set_mode(m_classifier, True)
    if not any(q.requires_grad for q in m_classifier.parameters()):
        # (the reference fails at loss.backward() with torch's "does not require grad" error in the same situation)
        raise RuntimeError("classifier_epoch_train: no parameter of the classifier requires grad (the vanilla classifiers freeze "
                           "themselves in train(): pass set_model_mode, as scripts/pretrain_classifier.py does)")
    trainer = make_surrogate_trainer(m_recipe, m_classifier)      # (built per epoch: which parts are frozen may have changed)
    _, n_ranks = distributed.world()
    reducer = distributed.GradBucketReducer(m_classifier.parameters()) if n_ranks > 1 else None
    prev = engine.precision_name()
    parts, total = [], 0
    try:
        for batch_idx, (_inputs, _targets) in enumerate(d_items):
            xs, zs = gen_input(_inputs, _targets)
            bits = engine.ones_mask_bits(xs.shape[0], n_players, xs.device)
            optimizer.zero_grad()
            _training.GRAD_SINK = reducer.ready if reducer is not None else None
            try:
                probs = trainer.forward_probs(xs, bits, train=True, seed=(seed or 0) + epoch)
                ce, dprobs = _cross_entropy(probs, zs.to(probs.device))   # CE applied on the soft-maxed output, as the reference does
                trainer.backward_probs(dprobs)
            finally:
                _training.GRAD_SINK = None
            if reducer is not None:
                reducer.finish()
            optimizer.step()
            hits = probs.argmax(dim=1).eq(zs.to(probs.device)).sum().float()
            parts.append(torch.stack([ce.reshape(()).float(), hits]))
            total += xs.shape[0]
    finally:
        engine.set_precision(prev)
    if not parts:
        return 0.0, 0.0
    cls_loss, correct = [float(v) for v in torch.stack(parts).sum(0).tolist()]
    env.log(f"  > epoch {epoch} :train // loss: cls {cls_loss / total:.6f} // acc: {100.0 * correct / total:.3f}%, {int(correct)}/{total}")
    return cls_loss / total, correct / total


def classifier_epoch_eval(env: Any, device: torch.device, n_players: int, d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe,
                          m_classifier, epoch: int, gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]]) -> Tuple[float, float]:
    """reference _classifier_epoch_eval (:153-189) -> (test_cls_loss, test_cls_acc)."""
    env = env or Log()
    m_classifier.eval()
    parts, total = [], 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, zs = gen_input(_inputs, _targets)
        ones = torch.ones((xs.shape[0], n_players), dtype=torch.long, device=device)
        with torch.no_grad():
            base, _ = m_recipe.fw_classifier(m_classifier, xs, ones)
        ce = _ce(base, zs.to(base.device))
        parts.append(torch.stack([ce.reshape(()).float(), base.argmax(dim=1).eq(zs.to(base.device)).sum().float()]))
        total += xs.shape[0]
    if not parts:
        return 0.0, 0.0
    cls_loss, correct = [float(v) for v in torch.stack(parts).sum(0).tolist()]
    env.log(f"  > epoch {epoch} :test // loss: cls {cls_loss / total:.6f} // acc: {100.0 * correct / total:.3f}%, {int(correct)}/{total}")
    return cls_loss / total, correct / total


def train_classifier(env: Any, device: torch.device, set_model_mode: Optional[Callable[[Any, bool], None]] = None) -> None:
    """reference train_classifier(env, device, set_model_mode=None) (:15-104).  ``env`` duck-typed as in
    scripts/train_explainer.train_explainer."""
    from ..utils.tools import set_iterative_seed
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env, save_epoch_ckpt_cfg
    env.log("[[[ train classifier ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.training.support_classifier:
        env.log("[[[ skip: classifier cannot be trained ]]]")
        return
    tcfg = config.train_classifier
    epoch_classifier, m_classifier = load_epoch_model_env(env, m_recipe, "classifier", device=device)
    if epoch_classifier >= tcfg.epochs:
        env.log("[[[ classifier already trained ]]]")
        return
    d_loader = load_cfg_dataset(env, getattr(config, "dataset", None))
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    n_players = m_recipe.n_players(m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    optimizer = torch.optim.AdamW(m_classifier.parameters(), lr=tcfg.lr, fused=True)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, tcfg.epochs)
    for epoch in range(epoch_classifier + 1, tcfg.epochs + 1):
        seed = set_iterative_seed(config.seed, f"train_classifier[epoch={epoch}]")
        env.log(f"### epoch {epoch}")
        if getattr(tcfg, "EXPERIMENTAL_progressive_training", None):      # trick for ltt (reference :51-56)
            freeze_lys = min(math.ceil(epoch / 1), m_config.num_hidden_layers)
            env.log(f"  > freeze side branches exc. first {freeze_lys} layers")
            m_classifier.ltt_freeze_layers_until(freeze_lys)
        ts_begin = time.time()
        tr = classifier_epoch_train(env, device, n_players, d_loader.train(tcfg.batch_size), m_recipe, m_classifier, set_model_mode,
                                    optimizer, epoch, gen_input, seed=seed)
        te = classifier_epoch_eval(env, device, n_players, d_loader.test(tcfg.batch_size), m_recipe, m_classifier, epoch, gen_input)
        scheduler.step()
        ts_delta = time.time() - ts_begin
        if hasattr(env, "metrics"):
            env.metrics({"epoch": epoch, "train_cls_loss": tr[0], "train_cls_acc": tr[1], "test_cls_loss": te[0], "test_cls_acc": te[1]})
        env.log(f"  > epoch {epoch} done in {ts_delta:.2f}s // train_loss: cls {tr[0]:.6f} // test_loss: cls {te[0]:.6f} // "
                f"test_acc: {te[1]:.3f}")
        if save_epoch_ckpt_cfg(env.model_path, "classifier", tcfg, epoch, m_classifier) and hasattr(env, "flush_cfg"):
            env.flush_cfg()
