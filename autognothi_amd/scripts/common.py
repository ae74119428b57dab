"""Small helpers shared by the pipeline bodies."""
from __future__ import annotations

from typing import Any, Callable, Optional

import torch


class Log:
    """Stand-in for the reference's ExpEnv (scripts/env.py:13): only ``.log`` is used by the loop bodies."""

    def __init__(self, sink: Optional[Callable[[str], None]] = None):
        self.sink = sink

    def log(self, msg: str) -> None:
        if self.sink is not None:
            self.sink(msg)


def device_rng(holder: Any, device: torch.device, seed: Optional[int]):
    """One device MT19937 per (holder, device); reseeded when `seed` is given (the per-epoch
    set_iterative_seed of scripts/train_explainer.py:64), else continuing its stream."""
    from .. import ops
    cache = holder.__dict__.setdefault("_ag_rng", {})
    key = str(device)
    if key not in cache:
        cache[key] = ops.DeviceMT19937(device, seed if seed is not None else 0)
    elif seed is not None:
        cache[key].seed(seed)
    return cache[key]
