"""The reference's stage machine (scripts/train_all.py:16-63): detect how far ``env.model_path`` has got from its checkpoints and
run what is left — base parameters -> classifier {0}, train classifier, classifier -> surrogate {0}, train surrogate, surrogate ->
explainer {0}, train explainer (train_explainer handles the duo recipes' extra head itself), everything -> final {0} after the coherency check (:166-218).
Checkpoints are the reference's wire format (scripts/resources.py:150-222), so a run can be continued by either implementation.
Out of scope here as in SURVEY §8: downloading pre-trained base parameters (params/loader.py) — stage 0 takes them from
``env.base_params`` (a state dict in the base model's key layout) or fails with a message saying so."""
from __future__ import annotations

import types
from typing import Any

import torch

from .resources import get_recipe, load_epoch_ckpt, load_epoch_model_env, save_epoch_ckpt
from .train_classifier import train_classifier
from .train_explainer import train_explainer
from .train_surrogate import train_surrogate

_STAGE0 = types.SimpleNamespace(epochs=0, ckpt_when="_:%1==0", lr=0.0, batch_size=1)   # reference Config_Train of conversions (:84-89)


def detect_stage(env: Any) -> int:
    """reference _detect_stage (:19-43): 0 nothing .. 7 final written."""
    config, path = env.config, env.model_path
    if load_epoch_ckpt(path, "final", 0)[0] is not None:
        return 7
    ep, _ = load_epoch_ckpt(path, "explainer", config.train_explainer.epochs)
    if ep is not None:
        return 6 if ep == config.train_explainer.epochs else 5
    ep, _ = load_epoch_ckpt(path, "surrogate", config.train_surrogate.epochs)
    if ep is not None:
        return 4 if ep == config.train_surrogate.epochs else 3
    ep, _ = load_epoch_ckpt(path, "classifier", 0)          # (sic: the reference looks for the epoch-0 conversion only)
    if ep is not None:
        return 2 if ep == config.train_classifier.epochs else 1
    return 0


def _fork(env: Any, section: str):
    """reference ``env.fork(lambda ec: ec.logger_*)`` (a per-stage logger); a duck-typed env without fork() is used as it is."""
    import contextlib
    if hasattr(env, "fork"):
        return env.fork(lambda ec: getattr(ec, "logger_" + section, None))
    return contextlib.nullcontext(env)


def train_all(env: Any, device: torch.device) -> None:
    stage = detect_stage(env)
    env.log(f"[[[ current stage: {stage} / 7 ]]]")
    if stage < 1:
        conv_pretrained_classifier(env)
    if stage < 2:
        with _fork(env, "classifier") as e_:
            train_classifier(e_, device)
    if stage < 3:
        conv_classifier_surrogate(env)
    if stage < 4:
        with _fork(env, "surrogate") as e_:
            train_surrogate(e_, device)
    if stage < 5:
        conv_surrogate_explainer(env)
    if stage < 6:
        with _fork(env, "explainer") as e_:
            train_explainer(e_, device)
    if stage < 7:
        conv_explainer_final(env, device)
    env.log("[[[ all stages ok ]]]")


def conv_pretrained_classifier(env: Any) -> None:
    """reference :66-104 (the tokenizer copy is the params loader's business: out of scope)."""
    base_params = getattr(env, "base_params", None)
    if base_params is None:
        raise NotImplementedError("stage 0 needs the pre-trained base parameters: attach them as env.base_params (state dict of "
                                  "config.net.base_model); fetching them (reference params/loader.py) is outside this build's scope")
    env.log("[[[ converting base -> classifier 0... ]]]")
    m_recipe, m_config = get_recipe(env.config)
    m_classifier = m_recipe.conv_pretrained_classifier(m_config, base_params)
    save_epoch_ckpt(env.model_path, "classifier", _STAGE0.ckpt_when, 0, 0, m_classifier)
    env.log("[[[ convert base -> classifier 0 ok ]]]")


def conv_classifier_surrogate(env: Any) -> None:
    """reference :107-121"""
    m_recipe, m_config = get_recipe(env.config)
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    epoch, m_classifier = load_epoch_model_env(env, m_recipe, "classifier")
    if epoch < env.config.train_classifier.epochs:
        raise ValueError("under-trained classifier")
    env.log(f"[[[ converting classifier {epoch} -> surrogate 0... ]]]")
    m_surrogate = m_recipe.conv_classifier_surrogate(m_config, m_misc, m_classifier)
    tc = env.config.train_surrogate
    save_epoch_ckpt(env.model_path, "surrogate", tc.ckpt_when, tc.epochs, 0, m_surrogate)
    env.log(f"[[[ convert classifier {epoch} -> surrogate 0 ok ]]]")


def conv_surrogate_explainer(env: Any) -> None:
    """reference :124-138"""
    m_recipe, m_config = get_recipe(env.config)
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    epoch, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate")
    if epoch < env.config.train_surrogate.epochs:
        raise ValueError("under-trained surrogate")
    env.log(f"[[[ converting surrogate {epoch} -> explainer 0... ]]]")
    m_explainer = m_recipe.conv_surrogate_explainer(m_config, m_misc, m_surrogate)
    tc = env.config.train_explainer
    save_epoch_ckpt(env.model_path, "explainer", tc.ckpt_when, tc.epochs, 0, m_explainer)
    env.log(f"[[[ convert surrogate {epoch} -> explainer 0 ok ]]]")


def conv_explainer_final(env: Any, device: torch.device) -> None:
    """reference :141-165; the coherency check (:168-218) runs on ``device`` (the reference runs it on the CPU; these modules
    compute on the GPU only)."""
    m_recipe, m_config = get_recipe(env.config)
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    ep_c, m_classifier = load_epoch_model_env(env, m_recipe, "classifier", device)
    ep_s, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate", device)
    ep_e, m_explainer = load_epoch_model_env(env, m_recipe, "explainer", device)
    for ep, sec in ((ep_c, "classifier"), (ep_s, "surrogate"), (ep_e, "explainer")):
        if ep < getattr(env.config, "train_" + sec).epochs:
            raise ValueError(f"under-trained {sec}")
    env.log("[[[ converting models -> final 0... ]]]")
    m_final = m_recipe.conv_explainer_final(m_config, m_misc, m_classifier, m_surrogate, m_explainer).to(device).eval()
    if not verify_final_coherency(env, device, m_recipe, m_config, m_misc, m_classifier, m_surrogate, m_explainer, m_final):
        raise ValueError("cannot save final model due to non-coherency")
    save_epoch_ckpt(env.model_path, "final", _STAGE0.ckpt_when, 0, 0, m_final)
    env.log("[[[ convert models -> final 0 ok ]]]")


def verify_final_coherency(env: Any, device: torch.device, m_recipe, m_config, m_misc, m_classifier, m_surrogate, m_explainer,
                           m_final, eps: float = 1e-5) -> bool:
    """reference _verify_final_coherency (:168-218): on the null input, the Final model's two outputs must equal the classifier's
    and the explainer's (given the surrogate's null value as grand and null) to 1e-5.  Checked in the fp32 parity mode."""
    from .. import engine
    env.log("[[[ verifying final model coherency... ]]]")
    if not m_recipe.measurements.verify_final_coherency:
        env.log("[[[ skipped: net recipe does not support this ]]]")
        return True
    n_players = m_recipe.n_players(m_config)
    nil_xs = m_recipe.gen_null(m_config, m_misc, device)
    nil_mask = torch.ones((1, n_players), dtype=torch.long, device=device)
    prev = engine.precision_name()
    engine.set_precision("fp32")
    try:
        with torch.no_grad():
            _, cls_ref = m_recipe.fw_classifier(m_classifier, nil_xs, nil_mask)
            srg_ref, _ = m_recipe.fw_surrogate(m_surrogate, nil_xs, nil_mask)
            exp_ref, _ = m_recipe.fw_explainer(m_explainer, nil_xs, nil_mask, srg_ref, srg_ref)
            cls_out, exp_out = m_recipe.fw_final(m_final, nil_xs)
    finally:
        engine.set_precision(prev)
    cls_diff = float((cls_ref - cls_out).abs().max())
    exp_diff = float((exp_ref - exp_out).abs().max())
    env.log(f"cls_diff: {cls_diff}, exp_diff: {exp_diff}")
    if cls_diff > eps or exp_diff > eps:
        env.log("[[[ !!! final is not coherent !!! ]]]")
        raise ValueError("final model is not coherent")
    env.log("[[[ verified final model is coherent ]]]")
    return True
