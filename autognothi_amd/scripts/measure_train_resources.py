"""Time and device memory of one surrogate / explainer training batch (reference scripts/measure_train_resources.py): the two batch
bodies (:176-262: forward under grad through the recipe's fw_* callables, the reference's loss, ``backward()``, no optimiser step
inside the measured region) run verbatim on the autograd bridge (autognothi_amd/autograd.py) with the drop-in models.shapley
functions; per-sample seconds and MB are collected until ``eval_train_resources.max_samples`` inputs have been seen.
Memory: the reference reads the largest single self-allocation out of a torch.profiler trace (:283-300, "wtf?" in the source); this
build's kernels allocate through the torch caching allocator without profiler events, so the figure here is the peak of
``torch.cuda.max_memory_allocated`` over the batch minus the level before it — the quantity that decides whether a batch fits."""
from __future__ import annotations

import gc
import time
from typing import Any, Callable, List, Optional, Tuple

import pydantic
import torch
from torch import Tensor

from ..models.shapley import loss_logits_kl_divergence, loss_shapley_new, mask_purely_uniform, mask_shapley_new
from ..recipes.types import ModelRecipe


class SecondsStats(pydantic.BaseModel):
    all: List[float]
    avg: float
    std: float

    @staticmethod
    def from_list(all: List[float]) -> "SecondsStats":
        t = torch.tensor(all, dtype=torch.float64)
        return SecondsStats(all=all, avg=float(t.mean()), std=float(t.std()) if len(all) > 1 else 0.0)


class MiBytesStats(SecondsStats):
    @staticmethod
    def from_list(all: List[float]) -> "MiBytesStats":
        t = torch.tensor(all, dtype=torch.float64)
        return MiBytesStats(all=all, avg=float(t.mean()), std=float(t.std()) if len(all) > 1 else 0.0)


class MeasureTrainResourcesReport(pydantic.BaseModel):
    """reference :53-59"""
    init_tm: float
    init_mem: float
    srg_tm: SecondsStats
    srg_mem: MiBytesStats
    exp_tm: SecondsStats
    exp_mem: MiBytesStats


def measure_props(func: Callable[[], Any], device: torch.device) -> Tuple[Any, float, float]:
    """reference _measure_props (:268-301): fn -> (result, seconds, MB)."""
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize(device)
    base = torch.cuda.memory_allocated(device)
    torch.cuda.reset_peak_memory_stats(device)
    t0 = time.perf_counter_ns()
    ret = func()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter_ns()
    return ret, (t1 - t0) / 1e9, max(torch.cuda.max_memory_allocated(device) - base, 0) / 1e6


def surrogate_batch_train(device: torch.device, n_players: int, m_recipe: ModelRecipe, m_classifier, m_surrogate,
                          optimizer: torch.optim.Optimizer, xs: Tensor) -> Tensor:
    """reference _surrogate_batch_train (:176-199), statement for statement."""
    b = xs.shape[0]
    mask_1 = torch.ones((b, n_players), dtype=torch.long, device=device)
    mask_rand = mask_purely_uniform(b, n_players).to(device)
    optimizer.zero_grad()
    m_classifier.eval()
    with torch.no_grad():
        _, orig = m_recipe.fw_classifier(m_classifier, xs, mask_1)
    optimizer.zero_grad()
    m_surrogate.train()
    adapt, _ = m_recipe.fw_surrogate(m_surrogate, xs, mask_rand)
    loss = loss_logits_kl_divergence(orig, adapt)
    loss.backward()
    return loss


def explainer_batch_train(device: torch.device, n_mask_samples: int, n_players: int, surrogate_null: Tensor, m_recipe: ModelRecipe,
                          m_surrogate, m_explainer, optimizer: torch.optim.Optimizer, xs: Tensor) -> Tensor:
    """reference _explainer_batch_train (:202-262); the K masks of an input share it (fw_surrogate accepts the un-expanded
    batch: INTEGRATION.md) instead of the reference's materialised Xs_EXT."""
    b = xs.shape[0]
    mask_1 = torch.ones((b, n_players), dtype=torch.long, device=device)
    mask_shap_ = mask_shapley_new(b * n_mask_samples, n_players).to(device)
    mask_shap = mask_shap_.reshape((b, n_mask_samples, n_players))
    optimizer.zero_grad()
    m_surrogate.eval()
    with torch.no_grad():
        v_s, _ = m_recipe.fw_surrogate(m_surrogate, xs, mask_shap_)
        v_1, _ = m_recipe.fw_surrogate(m_surrogate, xs, mask_1)
    optimizer.zero_grad()
    m_explainer.train()
    phi, _ = m_recipe.fw_explainer(m_explainer, xs, mask_1, v_1, surrogate_null)
    loss = loss_shapley_new(batch_size=b, n_mask_samples=n_mask_samples, n_players=n_players, mask=mask_shap, v_0=surrogate_null,
                            v_s=v_s, v_1=v_1, phi=phi)
    loss.backward()
    return loss


def measure_train_resources(env: Any, device: torch.device, d_loader: Optional[Any]) -> MeasureTrainResourcesReport:
    """reference measure_train_resources(env, device, d_loader) (:62-171).  Models are freshly constructed (random init), as there."""
    from .resources import get_recipe, load_cfg_dataset
    env.log("loading models...")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.measurements.allow_train_resources:
        raise ValueError("unsupported recipe action")
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    n_players = m_recipe.n_players(m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    nil_xs = m_recipe.gen_null(m_config, m_misc, device)
    if d_loader is None:
        env.log("loading dataset...")
        d_loader = load_cfg_dataset(env, getattr(getattr(config, "eval_performance", None), "dataset", None) or getattr(config, "dataset", None))

    def load_models():
        m_classifier = m_recipe.t_classifier(m_config).to(device=device)
        m_surrogate = m_recipe.t_surrogate(m_config).to(device=device)
        optim_srg = torch.optim.AdamW(m_surrogate.parameters(), lr=config.train_surrogate.lr)
        m_explainer = m_recipe.t_explainer(m_config).to(device=device)
        optim_exp = torch.optim.AdamW(m_explainer.parameters(), lr=config.train_explainer.lr)
        return m_classifier, m_surrogate, optim_srg, m_explainer, optim_exp

    (m_classifier, m_surrogate, optim_srg, m_explainer, optim_exp), init_tm, init_mem = measure_props(load_models, device)
    env.log(f"init: {init_tm:.6f} s, {init_mem:.2f} MB")
    batch_size, max_size = config.eval_train_resources.batch_size, config.eval_train_resources.max_samples
    m_surrogate.eval()
    with torch.no_grad():
        surrogate_null, _ = m_recipe.fw_surrogate(m_surrogate, nil_xs, torch.ones((1, n_players), dtype=torch.long, device=device))

    def sweep(step: Callable[[Tensor], Tensor], optimizer, tag: str) -> Tuple[List[float], List[float]]:
        seen, tms, mems = 0, [], []
        for _inputs, _targets in d_loader.train(batch_size):
            xs, _zs = gen_input(_inputs, _targets)
            size = xs.shape[0]
            _loss, tm, mem = measure_props(lambda: step(xs), device)
            optimizer.step()
            tms.append(tm / size)
            mems.append(mem / size)
            seen += size
            env.log(f"> {tag}: {tm / size:.6f} s, {mem / size:.2f} MB ({seen})")
            if seen >= max_size:
                break
        return tms, mems

    srg_tm, srg_mem = sweep(lambda xs: surrogate_batch_train(device, n_players, m_recipe, m_classifier, m_surrogate, optim_srg, xs),
                            optim_srg, "surrogate")
    exp_tm, exp_mem = sweep(lambda xs: explainer_batch_train(device, config.train_explainer.n_mask_samples, n_players, surrogate_null,
                                                             m_recipe, m_surrogate, m_explainer, optim_exp, xs), optim_exp, "explainer")
    return MeasureTrainResourcesReport(init_tm=init_tm, init_mem=init_mem, srg_tm=SecondsStats.from_list(srg_tm),
                                       srg_mem=MiBytesStats.from_list(srg_mem), exp_tm=SecondsStats.from_list(exp_tm),
                                       exp_mem=MiBytesStats.from_list(exp_mem))
