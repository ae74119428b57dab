"""Runtime / theoretical-work report on the HIP path (reference scripts/measure_performance.py:16-330): per-sample
latency of the classifier, surrogate, explainer and Final forwards (batch of one by default, synchronised around every
call as the reference's ``_measure_time``), parameter counts, and the GFLOP of one call.

The reference counts FLOPs with ``torch.profiler(with_flops=True)``; the kernels here are invisible to it, so the count
comes from the library's own per-launch accounting (``ag_profile_enable`` / ``ag_profile_collect``: 2 flops per MAC of
every GEMM and of the two attention contractions actually EXECUTED — e.g. a CLS-only last layer counts as such)."""
from __future__ import annotations

import ctypes as C
import gc
import time
from typing import Any, Callable, Iterable, List, Optional, Tuple

import pydantic
import torch
from torch import Tensor, nn

from .. import _lib as L
from ..recipes.types import ModelRecipe
from .common import Log

_CLASSES = (0, 1, 2, 3, 4, 5, 8, 9)


class ModelPerformance(pydantic.BaseModel):
    """reference :16-23 (seconds per sample, GFLOP per call, millions of parameters)"""
    time: List[float]
    time_avg: float
    time_std: float
    gflops: float
    params_all: float
    params_trainable: float


class MeasurePerformanceReport(pydantic.BaseModel):
    """reference :25-35"""
    classifier: Optional[ModelPerformance]
    surrogate: Optional[ModelPerformance]
    explainer: Optional[ModelPerformance]
    final: Optional[ModelPerformance]


def _sync() -> None:
    torch.cuda.synchronize()


def measure_time(func: Callable[[], Any]) -> Tuple[float, Any]:
    """reference _measure_time (:263-274) without emptying the allocator cache (that measures hipFree, not the model)."""
    with torch.no_grad():
        _sync()
        gc.collect()
        t0 = time.perf_counter_ns()
        ret = func()
        _sync()
        t1 = time.perf_counter_ns()
    return (t1 - t0) / 1e9, ret


def measure_flops(func: Callable[[], Any]) -> float:
    """FLOPs executed by one call, from the library's per-launch accounting."""
    lib = L.lib()
    _sync()
    L.check(lib.ag_profile_enable(1))
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    for c in _CLASSES:
        L.check(lib.ag_profile_collect(c, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
    with torch.no_grad():
        func()
    _sync()
    total = 0.0
    for c in _CLASSES:
        L.check(lib.ag_profile_collect(c, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
        total += fl.value
    L.check(lib.ag_profile_enable(0))
    return total


def stat_perf(model: nn.Module, tm: List[float], flops: float) -> ModelPerformance:
    """reference _stat_perf (:318-330)."""
    t = torch.tensor(tm, dtype=torch.float64)
    return ModelPerformance(
        time=tm, time_avg=float(t.mean()), time_std=float(t.std()) if len(tm) > 1 else 0.0, gflops=flops / 1e9,
        params_all=sum(p.numel() for p in model.parameters()) / 1e6,
        params_trainable=sum(p.numel() for p in model.parameters() if p.requires_grad) / 1e6)


def measure_performance(env: Any, device: torch.device, d_loader: Any) -> MeasurePerformanceReport:
    """reference measure_performance(env, device, d_loader) (:36-103): newest classifier / surrogate / explainer / final checkpoints
    of ``env.model_path``, one sample per inference (batch_size 1, as there), ``config.eval_performance.loops`` passes over the test
    split.  ``env`` duck-typed as in scripts/train_explainer.train_explainer; ``d_loader`` None falls back to ``env.d_loader``."""
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env
    env.log("loading models...")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if d_loader is None:
        env.log("loading dataset...")
        d_loader = load_cfg_dataset(env, getattr(config.eval_performance, "dataset", None) or getattr(config, "dataset", None))
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    meas = m_recipe.measurements
    models = {}
    for sec, need in (("classifier", meas.allow_performance_cls), ("surrogate", meas.allow_performance_srg_exp),
                      ("explainer", meas.allow_performance_srg_exp), ("final", meas.allow_performance_fin)):
        models[sec] = load_epoch_model_env(env, m_recipe, sec, device=device)[1] if need else None
    return measure_performance_loaded(env, device, m_recipe, m_recipe.n_players(m_config), lambda: d_loader.test(1),
                                      m_recipe.gen_input(m_config, m_misc, device), m_recipe.gen_null(m_config, m_misc, device),
                                      config.eval_performance.loops, m_classifier=models["classifier"], m_surrogate=models["surrogate"],
                                      m_explainer=models["explainer"], m_final=models["final"])


def measure_performance_loaded(env: Any, device: torch.device, m_recipe: ModelRecipe, n_players: int,
                               make_items: Callable[[], Iterable[Tuple[Any, Any]]], gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]],
                               null_xs: Tensor, loops: int, m_classifier=None, m_surrogate=None, m_explainer=None, m_final=None
                               ) -> MeasurePerformanceReport:
    """reference measure_performance (:38-103) on already-loaded models; any of the four may be None (its entry is None, as for a
    recipe whose allow_performance_* flag is off)."""
    env = env or Log()
    meas = m_recipe.measurements
    rep = {"classifier": None, "surrogate": None, "explainer": None, "final": None}

    def ones(size):
        return torch.ones((size, n_players), dtype=torch.long, device=device)

    if m_classifier is not None and meas.allow_performance_cls:
        m_classifier.eval()
        tm, xs = [], None
        for _ in range(loops):
            for _inputs, _targets in make_items():
                xs, zs = gen_input(_inputs, _targets)
                size = xs.shape[0]
                dt, _ = measure_time(lambda: m_recipe.fw_classifier(m_classifier, xs, ones(size)))
                tm.append(dt / size)
        fl = measure_flops(lambda: m_recipe.fw_classifier(m_classifier, xs, ones(xs.shape[0])))
        rep["classifier"] = stat_perf(m_classifier, tm, fl)
    if m_surrogate is not None and m_explainer is not None and meas.allow_performance_srg_exp:
        m_surrogate.eval(); m_explainer.eval()
        with torch.no_grad():
            surrogate_null, _ = m_recipe.fw_surrogate(m_surrogate, null_xs, ones(1))
        tm_s, tm_e, xs, grand = [], [], None, None
        for _ in range(loops):
            for _inputs, _targets in make_items():
                xs, zs = gen_input(_inputs, _targets)
                size = xs.shape[0]
                dt_s, out = measure_time(lambda: m_recipe.fw_surrogate(m_surrogate, xs, ones(size)))
                grand = out[0]
                dt_e, _ = measure_time(lambda: m_recipe.fw_explainer(m_explainer, xs, ones(size), grand, surrogate_null))
                tm_s.append(dt_s / size)
                tm_e.append(dt_e / size)
        fl_s = measure_flops(lambda: m_recipe.fw_surrogate(m_surrogate, xs, ones(xs.shape[0])))
        fl_e = measure_flops(lambda: m_recipe.fw_explainer(m_explainer, xs, ones(xs.shape[0]), grand, surrogate_null))
        rep["surrogate"] = stat_perf(m_surrogate, tm_s, fl_s)
        rep["explainer"] = stat_perf(m_explainer, tm_e, fl_e)
    if m_final is not None and meas.allow_performance_fin:
        m_final.eval()
        tm, xs = [], None
        for _ in range(loops):
            for _inputs, _targets in make_items():
                xs, zs = gen_input(_inputs, _targets)
                dt, _ = measure_time(lambda: m_recipe.fw_final(m_final, xs))
                tm.append(dt / xs.shape[0])
        fl = measure_flops(lambda: m_recipe.fw_final(m_final, xs))
        rep["final"] = stat_perf(m_final, tm, fl)
    for role, r in rep.items():
        if r is not None:
            env.log(f"PERFORMANCE RESULTS for {m_recipe.id} <{role[:3]}>: params all {r.params_all:.3f} M, trainable "
                    f"{r.params_trainable:.3f} M; flops {r.gflops:.3f} G; time mean {r.time_avg * 1e3:.3f} ms, std {r.time_std * 1e3:.3f} ms")
    return MeasurePerformanceReport(**rep)
