"""Experiment environment in the reference's on-disk format (reference scripts/env.py:13-127, scripts/types.py:260-296): a model
directory holds ``.hparams.json`` (the experiment config), ``.log.txt`` and the ``{section}-epoch-{n}.ckpt`` checkpoints
(scripts/resources.py).  ``ExpEnv(model_path)`` gives the pipeline entry points of this package (``train_all``, ``train_*``,
``measure_*``) the object the reference's scripts hand to theirs: ``.config`` / ``.model_path`` / ``.log`` / ``.metrics`` /
``.flush_cfg`` / ``.fork``, so a directory prepared for (or half-trained by) the reference can be continued here and back.
Out of scope as in SURVEY §8: Weights & Biases (no network: ``wandb_enabled`` is logged and ignored), the dataset loaders
(attach any object with ``.train(bs)`` / ``.test(bs)`` as ``d_loader``) and the download of pre-trained parameters (``base_params``)."""
from __future__ import annotations

import datetime
import json
import pathlib
from typing import Any, Callable, Dict, Optional

NET_VERSION = "beta.1.01"      # reference scripts/resources.py:79-82
_REQUIRED = ("seed", "dataset", "net", "train_classifier", "train_surrogate", "train_explainer", "eval_accuracy", "eval_faithfulness",
             "eval_cls_acc", "eval_performance", "eval_train_resources")
_TRAIN_DEFAULTS = {"EXPERIMENTAL_progressive_training": None}          # reference Config_Train optional field


class Cfg:
    """attribute view of a JSON object (nested objects become Cfg, lists stay lists); ``dump()`` gives the JSON back, in file
    order, with whatever the run has changed.  ``net.params`` stays a plain dict: recipes build their config from it."""

    def __init__(self, data: Dict[str, Any], raw_keys=()):
        for k, v in data.items():
            object.__setattr__(self, k, Cfg(v) if isinstance(v, dict) and k not in raw_keys else v)

    def dump(self) -> Dict[str, Any]:
        return {k: (v.dump() if isinstance(v, Cfg) else v) for k, v in self.__dict__.items() if not k.startswith("_ag_")}

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    def __repr__(self) -> str:
        return f"Cfg({self.dump()!r})"


def parse_config(data: Dict[str, Any]) -> Cfg:
    """validate the parts the pipelines rely on (the reference validates everything through pydantic ExpConfig) and build the view"""
    missing = [k for k in _REQUIRED if k not in data]
    if missing:
        raise ValueError(f".hparams.json: missing section(s) {missing}")
    net = data["net"]
    for k in ("kind", "version", "params"):
        if k not in net:
            raise ValueError(f".hparams.json: net.{k} missing")
    if net["version"] != NET_VERSION:
        raise ValueError(f"net version mismatch: expected {NET_VERSION}, got {net['version']}")     # (resources.py:79-82)
    for sec in ("train_classifier", "train_surrogate", "train_explainer"):
        for k in ("epochs", "ckpt_when", "lr", "batch_size"):
            if k not in data[sec]:
                raise ValueError(f".hparams.json: {sec}.{k} missing")
    if "n_mask_samples" not in data["train_explainer"]:
        raise ValueError(".hparams.json: train_explainer.n_mask_samples missing")
    cfg = Cfg({k: (Cfg(v, raw_keys=("params",)) if k == "net" else v) for k, v in data.items()})
    for sec in ("train_classifier", "train_surrogate", "train_explainer"):
        for k, dv in _TRAIN_DEFAULTS.items():
            if k not in getattr(cfg, sec):
                object.__setattr__(getattr(cfg, sec), "_ag_default_" + k, True)
                object.__setattr__(getattr(cfg, sec), k, dv)
    return cfg


def _dump_config(cfg: Cfg) -> Dict[str, Any]:
    out = cfg.dump()
    for sec in ("train_classifier", "train_surrogate", "train_explainer"):      # defaults this loader added are not written back
        view = getattr(cfg, sec)
        for k in _TRAIN_DEFAULTS:
            if view.__dict__.get("_ag_default_" + k) and out[sec].get(k) == _TRAIN_DEFAULTS[k]:
                out[sec].pop(k, None)
    return out


class ExpEnv:
    def __init__(self, model_path, d_loader: Any = None, base_params: Any = None, echo: bool = True, _forked=None):
        self.model_path = pathlib.Path(model_path)
        self.d_loader, self.base_params, self._echo = d_loader, base_params, echo
        if _forked is None:
            with open(self.model_path / ".hparams.json", "r", encoding="utf-8") as f:
                self.config = parse_config(json.load(f))
            self._log_fd = open(self.model_path / ".log.txt", "a", encoding="utf-8")
            self.log(f"[[[ NEW RUN: load config from {self.model_path.absolute().as_posix()} ]]]")
        else:
            self.config, self._log_fd = _forked

    def fork(self, get_logger_opts: Optional[Callable[[Any], Any]] = None) -> "ExpEnv":
        """reference :36-46 (a per-stage logger on the same config and log file)"""
        return ExpEnv(self.model_path, self.d_loader, self.base_params, self._echo, _forked=(self.config, self._log_fd))

    def log(self, msg: str) -> None:
        line = f"[{datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S.%f')}] {msg}"
        if self._echo:
            print(line, flush=True)
        if not self._log_fd.closed:
            self._log_fd.write(line + "\n")
            self._log_fd.flush()

    def metrics(self, data: Dict[str, Any]) -> None:
        """reference :74-88 without Weights & Biases: numbers and strings verbatim, everything else by type name"""
        self.log("METRICS: " + str({k: (v if isinstance(v, (float, int, str)) else f"<{type(v).__name__}>") for k, v in data.items()}))

    def flush_cfg(self) -> None:
        with open(self.model_path / ".hparams.json", "w", encoding="utf-8") as f:
            f.write(json.dumps(_dump_config(self.config), indent=2) + "\n")
        self.log("[i] updated config file")

    def __enter__(self) -> "ExpEnv":
        for sec in ("logger_classifier", "logger_surrogate", "logger_explainer"):
            opts = getattr(self.config, sec, None)
            if opts is not None and getattr(opts, "wandb_enabled", False):
                self.log(f"[[[ {sec}: wandb is not available in this build: metrics go to the log ]]]")
        return self

    def __exit__(self, *args) -> None:
        if not self._log_fd.closed:
            self._log_fd.flush()
