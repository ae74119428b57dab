"""Fine-tune a pre-trained classifier and export it as a base model (reference scripts/pretrain_classifier.py): the whole classifier is
unfrozen through ``train_classifier(..., set_model_mode=...)`` (:27-48), trained for ``config.train_classifier.epochs`` and written out
as ``<dest>/<model dir name>/{model.json, model.ckpt}`` (:60-67) — the layout the reference's params loader reads base models from.
``estimate_train_time`` (reference scripts/estimate_train_time.py) extrapolates the train_resources report to the configured run."""
from __future__ import annotations

import json
import pathlib
from typing import Any, Optional, Tuple

import torch

from ..utils.nnmodel import freeze_model_parameters
from .resources import get_recipe, load_epoch_ckpt, load_epoch_model_env
from .train_all import conv_pretrained_classifier
from .train_classifier import train_classifier


def pretrain_classifier(env: Any, device: torch.device, dest_root: Optional[pathlib.Path] = None) -> pathlib.Path:
    """reference pretrain_classifier(env, device) (:17-70) -> the export directory.  ``dest_root`` defaults to ``params/`` next to the
    model directory (the reference writes into its own package's ``params/``); a tokenizer (BERT misc) is the caller's to copy:
    tokenizers are outside this build's scope."""
    env.log("[[[ fine-tune pretrained model ]]]")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.training.support_classifier:
        raise ValueError("cannot fine-tune model: classification not supported")
    if config.net.kind not in ("vanilla_bert", "vanilla_vit"):
        raise ValueError(f"unsupported model kind: {config.net.kind}")

    def set_model_mode(net, _train):
        freeze_model_parameters(net, ..., requires_grad=True)

    epoch, _ = load_epoch_ckpt(env.model_path, "classifier", config.train_classifier.epochs)
    if epoch is None:
        env.log(":: initializing ft model")
        conv_pretrained_classifier(env)
        epoch = 0
    if epoch < config.train_classifier.epochs:
        env.log(f":: training ft model from epoch {epoch}")
        train_classifier(env, device, set_model_mode=set_model_mode)
    epoch, m_classifier = load_epoch_model_env(env, m_recipe, "classifier", device=device)
    if epoch < config.train_classifier.epochs:
        raise ValueError("classifier not fully trained")
    dest = pathlib.Path(dest_root if dest_root is not None else pathlib.Path(env.model_path).parent / "params") / pathlib.Path(env.model_path).name
    dest.mkdir(parents=True, exist_ok=True)
    params = config.net.params
    params = json.loads(params.model_dump_json()) if hasattr(params, "model_dump_json") else dict(params)
    with open(dest / "model.json", "w", encoding="utf-8") as f:
        f.write(json.dumps(params, indent=2))
    torch.save({k: v.detach().cpu() for k, v in m_classifier.state_dict().items()}, dest / "model.ckpt")
    env.log("[[[ fine-tuning complete ]]]")
    return dest


def fmt_tm(tm: float) -> str:
    """reference estimate_train_time.fmt_tm (:47-52)"""
    mins, hr = int(tm // 60) % 60, int(tm / 60 / 60)
    return f"     {mins:02d}m" if hr == 0 else f"{hr: 3d}h {mins:02d}m"


def estimate_train_time(env: Any, device: torch.device, train_size: Optional[int] = None) -> Tuple[float, float]:
    """reference estimate_train_time(env, device) (:13-44) -> (surrogate seconds, explainer seconds), from the (cached)
    train_resources report; ``train_size`` defaults to ``config.dataset.train_size`` (the reference prompts for it when absent)."""
    from .measure_all import load_or_run_report
    from .measure_train_resources import measure_train_resources
    env.log("[[[ retrieving training resource report... ]]]")
    config = env.config
    m_recipe, _ = get_recipe(config)
    if not m_recipe.measurements.allow_train_resources:
        env.log("[[[ error: cannot measure training speed ]]]")
        raise ValueError("given model does not support measurement")
    rep = load_or_run_report(env, "train_resources.json", lambda: measure_train_resources(env, device, None))
    if train_size is None:
        train_size = getattr(getattr(config, "dataset", None), "train_size", None)
    if train_size is None:
        raise ValueError("estimate_train_time: config.dataset.train_size is absent: pass train_size")
    e_c, e_s, e_e = config.train_classifier.epochs, config.train_surrogate.epochs, config.train_explainer.epochs
    tm_srg = rep["init_tm"] * (e_c + e_s) + rep["srg_tm"]["avg"] * train_size * (e_c + e_s)
    tm_exp = rep["init_tm"] * e_e + rep["exp_tm"]["avg"] * train_size * e_e
    env.log("[[[ estimated training time ]]]")
    env.log(f"> surrogate: {fmt_tm(tm_srg)}")
    env.log(f"> explainer: {fmt_tm(tm_exp)}")
    return tm_srg, tm_exp
