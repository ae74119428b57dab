"""Oracle (test infrastructure): mask samplers, Shapley loss / normalisation, KL loss,
faithfulness perturbation masks, iterative seeding.  numpy fp32; each function cites the
reference lines it restates."""
from __future__ import annotations

import hashlib
import random as _pyrandom
from typing import List, Optional, Tuple

import numpy as np

from .mt19937 import MT19937


# ----------------------------------------------------------------------------- seeding
def iterative_seed(master_seed: int, key: str) -> int:
    """reference utils/tools.py:46-54 (set_iterative_seed): sha256 -> first 8 bytes big-endian
    mod 2**32."""
    patt = f"[seed={master_seed},key={key}]"
    digest = hashlib.sha256(patt.encode("utf-8", "ignore")).digest()
    return int.from_bytes(digest[:8], byteorder="big") % 2 ** 32


# ----------------------------------------------------------------------------- samplers
def shapley_prefix_table(n_players: int) -> np.ndarray:
    """reference models/shapley.py:65-67 + :132 — size prior p(k) ∝ 1/(k(P-k)), k=1..P-1, and
    its exclusive prefix sum.  The reference computes it with torch.sum / torch.cumsum in fp32;
    their reduction order is a torch implementation detail, so the bit-exact table is a golden
    fixture (tests/golden/prefix_tables.npz) and this float64-then-round restatement is only
    checked against it to 1 ulp-level tolerance (tests/test_oracle_masks.py).  Mask parity
    tests always feed the fixture table."""
    k = np.arange(1, n_players, dtype=np.int64)
    w = (1.0 / (k * (n_players - k)).astype(np.float32)).astype(np.float32)
    p = (w / np.float32(w.astype(np.float64).sum())).astype(np.float32)
    return (np.cumsum(p.astype(np.float64)) - p).astype(np.float32)


def mask_shapley_new(n_mask_samples: int, n_players: int, gen: MT19937, prefix: np.ndarray) -> np.ndarray:
    """reference models/shapley.py:56-79 (+ _torch_choice :131-135).
    Draw order: U1 = rand(h, P) first (:69), then u2 = rand(h, 1) (:133)."""
    assert n_mask_samples % 2 == 0  # :62
    h = n_mask_samples // 2
    u1 = gen.rand_f32(h, n_players)
    u2 = gen.rand_f32(h, 1)
    pos = np.maximum((u2 >= prefix[None, :]).sum(axis=1) - 1, 0)  # :134
    thr = np.float32(1.0 / n_players) * pos.astype(np.float32)  # :70, fp32 product
    m = (u1 > thr[:, None]).astype(np.int64)  # :73
    return np.stack([m, 1 - m], axis=1).reshape(n_mask_samples, n_players)  # :76-78


def mask_purely_uniform(batch_size: int, n_features: int, gen: MT19937) -> np.ndarray:
    """reference models/shapley.py:109-115: (rand(B,P) > rand(B,1)).long(), drawn in that order."""
    a = gen.rand_f32(batch_size, n_features)
    b = gen.rand_f32(batch_size, 1)
    return (a > b).astype(np.int64)


def mask_uniform_selective(batch_size: int, n_features: int, n_masked: int, seed: Optional[int] = None) -> np.ndarray:
    """reference models/shapley.py:118-128 — python ``random.shuffle`` per row, first n_masked
    ids -> 0.  Uses the stdlib generator exactly as the reference does."""
    if seed is not None:
        _pyrandom.seed(seed)
    rows: List[List[int]] = []
    for _ in range(batch_size):
        ids = list(range(n_features))
        _pyrandom.shuffle(ids)
        off = set(ids[:n_masked])
        rows.append([0 if i in off else 1 for i in range(n_features)])
    return np.asarray(rows, dtype=np.int64).reshape(batch_size, n_features)


# ----------------------------------------------------------------------------- reductions
def normalize_shapley_explanation(pred: np.ndarray, grand: np.ndarray, null: np.ndarray) -> np.ndarray:
    """reference models/shapley.py:82-93.  pred [B,T,C] (T rows INCLUDE the CLS row),
    grand [B,C], null [1,C] -> pred + ((grand-null) - sum_t pred)/T."""
    pred = pred.astype(np.float32)
    t = pred.shape[1]
    diff = (grand[:, None, :] - null.reshape(1, 1, -1)).astype(np.float32) - pred.sum(axis=1, keepdims=True, dtype=np.float32)
    return (pred + diff / np.float32(t)).astype(np.float32)


def loss_shapley_new(batch_size: int, n_mask_samples: int, n_players: int, mask: np.ndarray,
                     v_0: np.ndarray, v_s: np.ndarray, v_1: np.ndarray, phi: np.ndarray
                     ) -> Tuple[np.float32, np.ndarray]:
    """reference models/shapley.py:9-53.  Returns (loss, dloss/dphi).
    v_hat[b,k,c] = v0[c] + sum_p mask[b,k,p] phi[b,c,p];  loss = P * mean((v_hat - v_s)^2)."""
    _ = v_1  # accepted and unused by the reference (:17)
    m = mask.reshape(batch_size, n_mask_samples, n_players).astype(np.float32)
    pred = v_0.reshape(1, 1, -1).astype(np.float32) + m @ phi.transpose(0, 2, 1).astype(np.float32)
    diff = pred.reshape(batch_size * n_mask_samples, -1) - v_s.astype(np.float32)
    n_el = diff.size
    loss = np.float32(n_players) * np.float32((diff.astype(np.float64) ** 2).sum() / n_el)
    d = diff.reshape(batch_size, n_mask_samples, -1)
    dphi = (2.0 * n_players / n_el) * np.einsum("bkp,bkc->bcp", m.astype(np.float64), d.astype(np.float64))
    return loss, dphi.astype(np.float32)


def _log_softmax(x: np.ndarray) -> np.ndarray:
    x = x - x.max(axis=-1, keepdims=True)
    return x - np.log(np.exp(x).sum(axis=-1, keepdims=True))


def loss_logits_kl_divergence(ref: np.ndarray, current: np.ndarray) -> np.float32:
    """reference models/shapley.py:96-106: F.kl_div(log_softmax(ref), softmax(current),
    'batchmean') = (1/B) sum t (log t - log_softmax(ref)), t = softmax(current).  Both inputs
    are already probabilities in the callers (scripts/train_surrogate.py:146) — kept."""
    ls = _log_softmax(ref.astype(np.float64))
    lt = _log_softmax(current.astype(np.float64))
    t = np.exp(lt)
    return np.float32((t * (lt - ls)).sum() / ref.shape[0])


# ----------------------------------------------------------------------------- faithfulness
def get_perturbed_samples(explanations: np.ndarray, n_players: int, steps: int, mask_base: int
                          ) -> Tuple[np.ndarray, np.ndarray]:
    """reference scripts/measure_faithfulness.py:225-251."""
    steps = min(n_players, steps)
    attribution = np.asarray(explanations).reshape(-1)
    ranking = np.argsort(attribution)[::-1]  # numpy default (quicksort/introsort) tie order, as the reference
    stops = np.linspace(0, n_players, steps, dtype=np.int64)
    masks = []
    for i in stops:
        m = np.ones((n_players,), dtype=np.int64) * mask_base
        m[ranking[:i]] ^= 1
        masks.append(m)
    return stops.astype(np.int64), np.asarray(masks, dtype=np.int64).reshape(len(stops), n_players)


def auc(values: np.ndarray) -> float:
    """reference scripts/measure_faithfulness.py:143-146."""
    v = np.asarray(values, dtype=np.float64)
    return float(((v[1:] + v[:-1]) / 2).mean())


def mc_permutation_shapley(probs_fn, perms: np.ndarray):
    """reference scripts/preview_text_shapley.py:62-153 restated.  probs_fn(masks [n, P] int64) -> surrogate outputs
    [n, C]; perms [reps, P].  -> (sv [C, P], v0 [C], vn [C])."""
    reps, p = perms.shape
    sv = None
    v0 = vn = None
    for r in range(reps):
        perm = perms[r]
        masks = np.zeros((p + 1, p), dtype=np.int64)
        for i in range(p + 1):
            masks[i, perm[:i]] = 1                                     # :87-90
        classes = np.asarray(probs_fn(masks), dtype=np.float32)
        e = np.exp(classes - classes.max(axis=1, keepdims=True))
        pr = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)      # :146  softmax of the (already soft-maxed) outputs
        vs = np.log(pr / (np.float32(1.0) - pr + np.float32(1e-6)))     # :148
        d_p = vs[1:] - vs[:-1]                                          # :118
        d = np.zeros_like(d_p)
        d[perm] = d_p                                                   # :122-123
        sv = d if sv is None else sv + d
        v0, vn = vs[0], vs[-1]                                          # :127-128
    return (sv.T.reshape(classes.shape[1], -1) / reps).astype(np.float32), v0, vn
