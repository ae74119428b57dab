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


def measure_accuracy(env: Any, device: torch.device, n_players: int, resolution: int,
                     make_items: Callable[[], Iterable[Tuple[Any, Any]]], m_recipe: ModelRecipe, m_surrogate, epoch: int,
                     gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]]) -> MeasureAccuracyReport:
    """reference measure_accuracy (:24-79): accuracy at ``resolution`` masked-player counts from 0 to P."""
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
