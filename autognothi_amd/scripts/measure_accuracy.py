"""Accuracy reports on the HIP path (reference scripts/measure_accuracy.py:24-110, scripts/measure_cls_acc.py:106-128):
surrogate accuracy under a fixed number of masked players, and the Final model's classification accuracy.
The exact-cardinality sampler stays on python ``random`` exactly as the reference (models/shapley.py:118-128 — host
stdlib, a few hundred ints per batch); the forwards are the device path."""
from __future__ import annotations

import random
import time
from typing import Any, Callable, Iterable, List, Optional, Tuple

import pydantic
import torch
from torch import Tensor

from .. import ops
from ..recipes.types import ModelRecipe
from .common import Log


class MeasureAccuracyReport(pydantic.BaseModel):
    """reference scripts/measure_accuracy.py:15-23"""
    masked_players: List[int]
    accuracy: List[float]


class MeasureClsAccReport(pydantic.BaseModel):
    """reference scripts/measure_cls_acc.py (epochs x accuracy of the Final model's classifier output)"""
    epochs: List[int]
    accuracy: List[float]


def mask_uniform_selective(batch_size: int, n_features: int, n_masked: int) -> Tensor:
    """reference models/shapley.py:118-128 (python ``random.shuffle`` per row; first n_masked ids -> 0)."""
    ret: List[List[int]] = []
    for _ in range(batch_size):
        ids = list(range(n_features))
        random.shuffle(ids)
        off = set(ids[:n_masked])
        ret.append([0 if i in off else 1 for i in range(n_features)])
    return torch.tensor(ret, dtype=torch.long)


def measure_surrogate_epoch(env: Any, device: torch.device, n_players: int, n_masked_players: int,
                            d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_surrogate, epoch: int,
                            gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]]) -> float:
    """reference _measure_surrogate_epoch (:82-110) -> accuracy in [0, 1]."""
    env = env or Log()
    correct, total = 0, 0
    m_surrogate.eval()
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, zs = gen_input(_inputs, _targets)
        b = xs.shape[0]
        bits = ops.pack_mask(mask_uniform_selective(b, n_players, n_masked_players).to(device))
        with torch.no_grad():
            adapt, _ = m_recipe.fw_surrogate(m_surrogate, xs, bits)
        correct += int(adapt.argmax(dim=1).eq(zs.to(adapt.device)).sum().item())
        total += b
        env.log(f"  > mask {n_masked_players} :{batch_idx}:test // acc: {100.0 * correct / total:.3f}%, {correct}/{total}")
    return correct / max(total, 1)


def measure_accuracy_loaded(env: Any, device: torch.device, n_players: int, resolution: int,
                             make_items: Callable[[], Iterable[Tuple[Any, Any]]], m_recipe: ModelRecipe, m_surrogate, epoch: int,
                             gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]]) -> MeasureAccuracyReport:
    """reference measure_accuracy (:48-79) given a loaded surrogate: accuracy at ``resolution`` masked-player counts 0..P."""
    env = env or Log()
    if not m_recipe.measurements.allow_accuracy:
        raise ValueError("unsupported recipe action")
    all_masked = torch.linspace(0, n_players, resolution, dtype=torch.long).tolist()
    accs: List[float] = []
    for n_masked in all_masked:
        t0 = time.time()
        acc = measure_surrogate_epoch(env, device, n_players, int(n_masked), make_items(), m_recipe, m_surrogate, epoch, gen_input)
        accs.append(acc)
        env.log(f"  > mask {n_masked} done in {time.time() - t0:.2f}s // test_acc: {acc:.3f}")
    return MeasureAccuracyReport(masked_players=[int(x) for x in all_masked], accuracy=accs)


def measure_final_cls_epoch(env: Any, d_items: Iterable[Tuple[Any, Any]], m_recipe: ModelRecipe, m_final, epoch: int,
                            gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]]) -> float:
    """reference _measure_final_cls_epoch (scripts/measure_cls_acc.py:106-128)."""
    env = env or Log()
    correct, total = 0, 0
    m_final.eval()
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        xs, zs = gen_input(_inputs, _targets)
        with torch.no_grad():
            fin, _ = m_recipe.fw_final(m_final, xs)
        correct += int(fin.argmax(dim=1).eq(zs.to(fin.device)).sum().item())
        total += xs.shape[0]
        env.log(f"  > epoch {epoch} :{batch_idx}:test // acc: {100.0 * correct / total:.3f}%, {correct}/{total}")
    return correct / max(total, 1)


def measure_accuracy(env: Any, device: torch.device, d_loader: Optional[Any]) -> MeasureAccuracyReport:
    """reference measure_accuracy(env, device, d_loader) (:26-79): newest surrogate checkpoint of ``env.model_path``,
    ``config.eval_accuracy.resolution`` masked-player counts, test batches of ``config.train_surrogate.batch_size``.
    ``env`` duck-typed as in scripts/train_explainer.train_explainer; ``d_loader`` None falls back to ``env.d_loader``."""
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env
    env.log("[[[ measuring model accuracy ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.measurements.allow_accuracy:
        raise ValueError("unsupported recipe action")
    if d_loader is None:
        env.log("loading dataset...")
        d_loader = load_cfg_dataset(env, getattr(config.eval_accuracy, "dataset", None) or getattr(config, "dataset", None))
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    n_players = m_recipe.n_players(m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    epoch_surrogate, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate", device=device)
    env.log("[[[ measuring surrogate... ]]]")
    return measure_accuracy_loaded(env, device, n_players, config.eval_accuracy.resolution,
                                   lambda: d_loader.test(config.train_surrogate.batch_size), m_recipe, m_surrogate,
                                   epoch_surrogate, gen_input)


def measure_cls_acc(env: Any, device: torch.device, d_loader: Optional[Any]) -> MeasureClsAccReport:
    """reference measure_cls_acc(env, device, d_loader) (scripts/measure_cls_acc.py:32-103): for every explainer checkpoint
    selected by ``config.eval_cls_acc.on_exp_epochs`` (None: the last training epoch only) assemble the Final model from the
    newest classifier + surrogate and that explainer (conv_explainer_final) and measure its classification accuracy on the
    test split in batches of ``config.train_classifier.batch_size``."""
    from .resources import get_epoch_ckpts, get_recipe, load_cfg_dataset, load_epoch_ckpt, load_epoch_model_env, ranged_modulo_test
    env.log("[[[ measuring classifier accuracy ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.measurements.allow_cls_acc:
        raise ValueError("unsupported recipe action")
    if d_loader is None:
        env.log("loading dataset...")
        d_loader = load_cfg_dataset(env, getattr(config.eval_cls_acc, "dataset", None) or getattr(config, "dataset", None))
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    _, m_classifier = load_epoch_model_env(env, m_recipe, "classifier", device=device)
    _, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate", device=device)
    pattern = config.eval_cls_acc.on_exp_epochs

    def measure_on(ep: int) -> bool:
        return ep == config.train_explainer.epochs if pattern is None else ranged_modulo_test(pattern)(ep)

    env.log("[[[ measuring explainers... ]]]")
    epochs: List[int] = []
    accs: List[float] = []
    for ep in get_epoch_ckpts(env.model_path, "explainer", config.train_explainer.epochs):
        if not measure_on(ep):
            continue
        epoch_explainer, sd = load_epoch_ckpt(env.model_path, "explainer", ep, required=True)
        m_explainer = m_recipe.t_explainer(m_config)
        m_explainer.load_state_dict(sd)
        m_final = m_recipe.conv_explainer_final(m_config, m_misc, m_classifier, m_surrogate, m_explainer.to(device)).to(device)
        t0 = time.time()
        acc = measure_final_cls_epoch(env, d_loader.test(config.train_classifier.batch_size), m_recipe, m_final, epoch_explainer, gen_input)
        epochs.append(epoch_explainer)
        accs.append(acc)
        env.log(f"  > epoch {epoch_explainer} done in {time.time() - t0:.2f}s // test_acc: {acc:.3f}")
    return MeasureClsAccReport(epochs=epochs, accuracy=accs)
