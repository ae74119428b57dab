"""Seeding helpers mirrored from reference utils/tools.py:33-54."""
from __future__ import annotations

import hashlib
import os
import random

import numpy as np
import torch


def derive_seed(master_seed: int, key: str) -> int:
    """sha256("[seed=..,key=..]")[:8] big-endian mod 2**32 (reference utils/tools.py:50-52)."""
    digest = hashlib.sha256(f"[seed={master_seed},key={key}]".encode("utf-8", "ignore")).digest()
    return int.from_bytes(digest[:8], byteorder="big") % 2 ** 32


def set_seed(seed: int) -> None:
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def set_iterative_seed(master_seed: int, key: str) -> int:
    """Reseed every global generator from (master, key) so a resumed run replays the same masks
    (scripts/train_explainer.py:64).  Returns the derived 32-bit seed (what the device sampler is seeded with)."""
    seed = derive_seed(master_seed, key)
    set_seed(seed)
    return seed
