"""Deterministic synthetic weights / inputs shared by the golden-fixture generator, the
parity tests and bench.py.

There is no network on the build or GPU boxes, so no pretrained checkpoints: every run uses
seeded random weights.  Constructor-order random init cannot be reproduced across two model
implementations, so weights are keyed by *parameter name* instead:
the reference model (in the fixture generator) and this package's model (on the GPU box) get
bit-identical tensors through ``load_state_dict`` because their state-dict key sets are
identical by contract.

numpy's PCG64 stream is stable across numpy versions, torch's CPU randn is not guaranteed to
be; hence numpy.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode("utf-8")), seed & 0xFFFFFFFF])


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """fp32 tensor for state-dict key `name`.

    Scales are chosen so that a random-init transformer has O(1) activations and a
    non-degenerate soft-max (so masks visibly move the outputs and parity tests bite)."""
    g = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    low = name.lower()
    if name == "surrogate_null":
        return np.zeros(shape, dtype=np.float32)
    if len(shape) == 1:
        if leaf == "weight":  # 1-D weights are LayerNorm gains everywhere on this path
            return (1.0 + 0.05 * g.standard_normal(shape)).astype(np.float32)
        return (0.05 * g.standard_normal(shape)).astype(np.float32)
    if "embeddings" in low and len(shape) == 2:  # BERT word/pos/type tables
        return (0.5 * g.standard_normal(shape)).astype(np.float32)
    if leaf in ("cls_token", "position_embeddings"):
        return (0.3 * g.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    gain = 2.0 if (".classifier." in "." + name or name.startswith("classifier.")) else 1.0
    return (gain / np.sqrt(fan_in) * g.standard_normal(shape)).astype(np.float32)


def synth_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, np.ndarray]:
    return {k: synth_tensor(k, s, seed) for k, s in shapes}


def load_synth_weights(module, seed: int = 0) -> None:
    """Fill every tensor of ``module.state_dict()`` (torch nn.Module — the reference's or
    ours) with :func:`synth_tensor` of its key, via load_state_dict."""
    import torch

    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            new[k] = torch.from_numpy(synth_tensor(k, tuple(v.shape), seed)).to(v.dtype)
        else:
            new[k] = v
    module.load_state_dict(new)


def synth_images(batch: int, px: int = 224, channels: int = 3, seed: int = 0) -> np.ndarray:
    """~N(0,1) pixels, like normalised images (reference datasets/loader.py:388-389)."""
    g = np.random.default_rng([0x1A6E, seed & 0xFFFFFFFF])
    return g.standard_normal((batch, channels, px, px)).astype(np.float32)


def synth_token_ids(batch: int, seq_len: int = 128, vocab: int = 30522, seed: int = 0) -> np.ndarray:
    """ids in [1000, min(30000, vocab)), ids[:,0]=101 ([CLS])."""
    g = np.random.default_rng([0x70C5, seed & 0xFFFFFFFF])
    hi = min(30000, vocab)
    lo = min(1000, hi - 1)
    ids = g.integers(lo, hi, size=(batch, seq_len), dtype=np.int64)
    ids[:, 0] = min(101, vocab - 1)
    return ids


def synth_null_ids(seq_len: int = 128, vocab: int = 30522) -> np.ndarray:
    """Stand-in for tokenizer("") padded to seq_len: [CLS]=101, [SEP]=102, [PAD]=0...
    (reference recipes/vanilla_bert.py:265-278 needs a tokenizer, absent offline)."""
    ids = np.zeros((1, seq_len), dtype=np.int64)
    ids[0, 0] = min(101, vocab - 1)
    ids[0, 1] = min(102, vocab - 1)
    return ids
