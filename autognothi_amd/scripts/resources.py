"""Checkpoint wire format of the reference (scripts/resources.py:150-271): ``{section}-epoch-{n}.ckpt`` files, each a
plain ``OrderedDict`` state dict written by ``torch.save``.  The modules of this package keep the reference's state-dict
keys and shapes (tests/golden/state_keys*.json), so checkpoints trained by the reference load into the HIP-backed modules
and checkpoints written here load into the reference, unchanged."""
from __future__ import annotations

import os
import pathlib
import re
from collections import OrderedDict
from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import nn

SECTIONS = ("classifier", "surrogate", "explainer", "final")


def ranged_modulo_test(pattern: str) -> Callable[[int], bool]:
    """``"<=10:%2==1; _:%10==0"`` -> predicate on the epoch number (reference utils/strings.py:119-151): each clause
    covers the epochs after the previous clause's bound up to its own (``_`` = unbounded)."""
    clauses: List[Tuple[int, int, int]] = []
    for part in (x.strip() for x in pattern.split(";")):
        if not part:
            continue
        m1 = re.findall(r"<=\s*(\d+)\s*:\s*%\s*(\d+)\s*==\s*(\d+)", part)
        m2 = re.findall(r"_\s*:\s*%\s*(\d+)\s*==\s*(\d+)", part)
        if m1:
            bnd, mod, rem = map(int, m1[0])
        elif m2:
            bnd, (mod, rem) = 10 ** 9, map(int, m2[0])
        else:
            raise ValueError(f"invalid pattern: {pattern}")
        clauses.append((bnd, mod, rem))
    clauses.sort(key=lambda c: c[0])
    ranges, low = [], 0
    for bnd, mod, rem in clauses:
        ranges.append((low, bnd, mod, rem))
        low = bnd + 1

    def test(num: int) -> bool:
        return any(lo <= num <= hi and num % mod == rem for lo, hi, mod, rem in ranges)
    return test


def ckpt_path(path: pathlib.Path, section: str, epoch: int) -> pathlib.Path:
    return pathlib.Path(path) / f"{section}-epoch-{epoch}.ckpt"


def load_epoch_ckpt(path: pathlib.Path, section: str, max_epochs: int, required: bool = False
                    ) -> Tuple[Optional[int], Optional["OrderedDict[str, torch.Tensor]"]]:
    """Newest ``{section}-epoch-{n}.ckpt`` with n <= max_epochs (reference :150-169) -> (epoch, state dict on the CPU)."""
    path = pathlib.Path(path)
    files = {p.name for p in path.iterdir()}
    for epoch in range(max_epochs, -1, -1):
        name = f"{section}-epoch-{epoch}.ckpt"
        if name in files:
            return epoch, torch.load(path / name, weights_only=False, map_location=torch.device("cpu"))
    if required:
        raise FileNotFoundError(f"no checkpoint found for '{section}' under '{path}'")
    return None, None


def get_epoch_ckpts(path: pathlib.Path, section: str, max_epochs: int) -> List[int]:
    return [e for e in range(max_epochs + 1) if ckpt_path(path, section, e).exists()]


def save_epoch_ckpt(path: pathlib.Path, section: str, ckpt_when: str, epochs: int, epoch: int,
                    state_dict: Union[nn.Module, "OrderedDict[str, torch.Tensor]"]) -> bool:
    """reference :182-222: always write this epoch; drop the previous epoch's file unless it is the initial, a scheduled
    (``ckpt_when``) or the final checkpoint."""
    keep = ranged_modulo_test(ckpt_when)

    def should_keep(ep: int) -> bool:
        return ep == 0 or keep(ep) or ep == epochs
    if isinstance(state_dict, nn.Module):
        state_dict = OrderedDict((k, v.detach().cpu()) for k, v in state_dict.state_dict().items())
    this = ckpt_path(path, section, epoch)
    tmp = this.with_name(this.name + ".tmp")      # written aside and renamed: a reader (or a crash) never sees half a file
    torch.save(state_dict, tmp)
    os.replace(tmp, this)
    if not should_keep(epoch - 1):
        ckpt_path(path, section, epoch - 1).unlink(missing_ok=True)
    return True


def save_epoch_ckpt_main(path: pathlib.Path, section: str, cfg, epoch: int, state_dict, env=None) -> bool:
    """the pipelines' end-of-epoch write with one process per GPU: rank 0 writes the checkpoint (parameters are replicated) and
    flushes the config, every rank then waits at a barrier so that none resumes from — or deletes — a file still being
    written.  One rank: ``save_epoch_ckpt_cfg`` + ``env.flush_cfg()``, as the reference (scripts/train_explainer.py:119-123)."""
    from .. import distributed
    saved = True
    if distributed.is_main():
        saved = save_epoch_ckpt_cfg(path, section, cfg, epoch, state_dict)
        if saved and env is not None and hasattr(env, "flush_cfg"):
            env.flush_cfg()
    distributed.barrier()
    return saved


def get_recipe(config) -> Tuple[object, object]:
    """reference get_recipe (:55-83): experiment config -> (recipe, model config).  ``config.net.kind`` picks the recipe;
    ``config.net.params`` may be this package's config object, the reference's pydantic model or a plain dict."""
    from ..recipes import get_recipe as by_kind
    recipe = by_kind(config.net.kind)
    params = config.net.params
    if isinstance(params, recipe.t_config):
        return recipe, params
    if hasattr(params, "model_dump"):
        params = params.model_dump()
    return recipe, recipe.t_config(**dict(params))


def load_cfg_dataset(env, d_config=None):
    """reference load_cfg_dataset (:86-147) builds a DatasetLoader from ``config.dataset``; dataset loading is outside this
    build's scope (datasets/params loaders), so the loader is taken from the environment: any object with
    ``.train(batch_size)`` / ``.test(batch_size)`` iterables of ``(_inputs, _targets)`` that ``recipe.gen_input`` accepts
    (the reference's own ``datasets.loader.DatasetLoader`` qualifies)."""
    loader = getattr(env, "d_loader", None)
    if loader is None:
        raise NotImplementedError("attach a dataset loader as env.d_loader (.train(bs) / .test(bs)); the reference's "
                                  "datasets/loader.py is out of this build's scope")
    return loader


def load_epoch_model_env(env, m_recipe, section: str, device: torch.device = torch.device("cpu")) -> Tuple[int, nn.Module]:
    """reference load_epoch_model(env, m_recipe, section, device) (:225-271)."""
    _, m_config = get_recipe(env.config)
    max_epochs = 0 if section == "final" else getattr(env.config, "train_" + section).epochs
    return load_epoch_model(env.model_path, m_recipe, m_config, section, max_epochs, device)


def save_epoch_ckpt_cfg(path: pathlib.Path, section: str, cfg, epoch: int, state_dict) -> bool:
    """reference save_epoch_ckpt(path, id, cfg: Config_Train, epoch, state_dict) (:182-222)."""
    return save_epoch_ckpt(path, section, cfg.ckpt_when, cfg.epochs, epoch, state_dict)


def load_epoch_model(model_path: pathlib.Path, m_recipe, m_config, section: str, max_epochs: int,
                     device: torch.device = torch.device("cpu")) -> Tuple[int, nn.Module]:
    """reference load_epoch_model (:225-271): build ``t_{section}(config)``, load the newest checkpoint, ``.eval()``."""
    if section not in SECTIONS:
        raise ValueError(f"unknown section {section}")
    epoch, sd = load_epoch_ckpt(model_path, section, 0 if section == "final" else max_epochs, required=True)
    model = getattr(m_recipe, "t_" + section)(m_config)
    model.load_state_dict(sd)
    return epoch, model.to(device=device).eval()
