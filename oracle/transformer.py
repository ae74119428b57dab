"""Oracle (test infrastructure): fp32 numpy restatement of the reference's masked ViT / BERT
forward, classifier/surrogate heads and explainer heads.  Weights come in as a dict of numpy
arrays keyed by the reference's state-dict names."""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
from scipy.special import erf as _erf

from .shapley import normalize_shapley_explanation

SD = Dict[str, np.ndarray]
F32_MIN = np.float32(np.finfo(np.float32).min)


def linear(x: np.ndarray, sd: SD, prefix: str) -> np.ndarray:
    """torch.nn.Linear: x @ W^T + b, W stored [out, in]."""
    return (x @ sd[prefix + ".weight"].T + sd[prefix + ".bias"]).astype(np.float32)


def layer_norm(x: np.ndarray, sd: SD, prefix: str, eps: float) -> np.ndarray:
    x64 = x.astype(np.float64)
    mu = x64.mean(axis=-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(axis=-1, keepdims=True)
    y = (x64 - mu) / np.sqrt(var + eps)
    return (y * sd[prefix + ".weight"] + sd[prefix + ".bias"]).astype(np.float32)


def gelu(x: np.ndarray) -> np.ndarray:
    """nn.GELU() default = exact erf form (reference models/vanilla_vit.py:491)."""
    return (0.5 * x * (1.0 + _erf(x.astype(np.float64) / math.sqrt(2.0)))).astype(np.float32)


def softmax(x: np.ndarray) -> np.ndarray:
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def _heads(x: np.ndarray, nh: int) -> np.ndarray:
    r, t, h = x.shape
    return x.reshape(r, t, nh, h // nh).transpose(0, 2, 1, 3)


def self_attention(u: np.ndarray, mask: np.ndarray, sd: SD, prefix: str, nh: int, mode: str) -> np.ndarray:
    """reference models/vanilla_vit.py:436-465 (mode 'vit': scores *= mask on the key axis) and
    models/vanilla_bert.py:503-537 (mode 'bert': scores += (1-mask)*finfo(f32).min, the HF
    extended mask of :264-266)."""
    q = _heads(linear(u, sd, prefix + ".query"), nh)
    k = _heads(linear(u, sd, prefix + ".key"), nh)
    v = _heads(linear(u, sd, prefix + ".value"), nh)
    d = q.shape[-1]
    s = (q @ k.transpose(0, 1, 3, 2)).astype(np.float32) / np.float32(math.sqrt(d))
    m = mask.astype(np.float32)[:, None, None, :]
    if mode == "vit":
        s = s * m
    else:
        s = s + (np.float32(1.0) - m) * F32_MIN
    p = softmax(s)
    ctx = (p @ v).astype(np.float32)
    r, _, t, _ = ctx.shape
    return ctx.transpose(0, 2, 1, 3).reshape(r, t, nh * d)


def vit_layer(h: np.ndarray, mask: np.ndarray, sd: SD, prefix: str, nh: int, eps: float,
              norm1_identity: bool = False) -> np.ndarray:
    """reference models/vanilla_vit.py:364-377 (pre-LN block)."""
    u = h if norm1_identity else layer_norm(h, sd, prefix + ".layernorm_before", eps)
    a = self_attention(u, mask, sd, prefix + ".attention.self", nh, "vit")
    h = h + linear(a, sd, prefix + ".attention.output.dense")
    w = layer_norm(h, sd, prefix + ".layernorm_after", eps)
    inter = gelu(linear(w, sd, prefix + ".intermediate.dense"))
    return (linear(inter, sd, prefix + ".output.dense") + h).astype(np.float32)


def vit_embeddings(x: np.ndarray, sd: SD, prefix: str, patch: int) -> np.ndarray:
    """reference models/vanilla_vit.py:242-253, :279-284: Conv2d(k=s=patch) == per-patch GEMM
    with the (c, ph, pw) flattening of weight.view(H, C*patch*patch)."""
    b, c, hh, ww = x.shape
    gh, gw = hh // patch, ww // patch
    w = sd[prefix + ".patch_embeddings.projection.weight"]
    hdim = w.shape[0]
    patches = x.reshape(b, c, gh, patch, gw, patch).transpose(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * patch * patch)
    e = patches @ w.reshape(hdim, -1).T + sd[prefix + ".patch_embeddings.projection.bias"]
    cls = np.broadcast_to(sd[prefix + ".cls_token"], (b, 1, hdim))
    return (np.concatenate([cls, e], axis=1) + sd[prefix + ".position_embeddings"]).astype(np.float32)


def vit_model(x: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict, prefix: str = "vit",
              collect: Optional[list] = None) -> np.ndarray:
    """reference models/vanilla_vit.py:207-214.  mask_t is the T-wide mask (CLS column already
    prepended, recipes/vanilla_vit.py:219-224)."""
    h = vit_embeddings(x, sd, prefix + ".embeddings", cfg["img_patch_size"])
    if collect is not None:
        collect.append(h)
    for i in range(cfg["num_hidden_layers"]):
        h = vit_layer(h, mask_t, sd, f"{prefix}.encoder.layers.{i}", cfg["num_attention_heads"], cfg["layer_norm_eps"])
        if collect is not None:
            collect.append(h)
    return layer_norm(h, sd, prefix + ".layernorm", cfg["layer_norm_eps"])


def prepend_cls(mask: np.ndarray) -> np.ndarray:
    """recipes/vanilla_vit.py:219-224 / recipes/vanilla_bert.py:281-290."""
    return np.concatenate([np.ones((mask.shape[0], 1), dtype=mask.dtype), mask], axis=1)


def vit_surrogate(x: np.ndarray, mask_p: np.ndarray, sd: SD, cfg: dict, collect: Optional[list] = None) -> np.ndarray:
    """fw_surrogate / fw_classifier for vanilla ViT: probabilities [R, C]
    (models/vanilla_vit.py:51-56)."""
    z = vit_model(x, prepend_cls(mask_p), sd, cfg, "vit", collect)
    return softmax(linear(z[:, 0, :], sd, "classifier"))


def _vit_explainer_head(z: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict,
                        grand: Optional[np.ndarray], null: Optional[np.ndarray]) -> np.ndarray:
    o = z
    for j in range(cfg["explainer_attn_num_layers"]):
        o = vit_layer(o, mask_t, sd, f"explainer_attn.{j}", cfg["num_attention_heads"], cfg["layer_norm_eps"],
                      norm1_identity=(j == 0))
    o = layer_norm(o, sd, "explainer_mlp.0", 1e-5)  # torch default eps (models/vanilla_vit.py:94)
    o = gelu(linear(o, sd, "explainer_mlp.1"))
    o = gelu(linear(o, sd, "explainer_mlp.3"))
    o = linear(o, sd, "explainer_mlp.5")
    if cfg["explainer_normalize"]:
        o = normalize_shapley_explanation(o, grand, null)
    return np.ascontiguousarray(o[:, 1:, :].transpose(0, 2, 1))


def vit_explainer(x: np.ndarray, mask_p: np.ndarray, grand: np.ndarray, null: np.ndarray, sd: SD, cfg: dict,
                  duo: bool = False):
    """fw_explainer for vanilla / froyo ViT (models/vanilla_vit.py:102-130); duo=True follows
    models/duo_vanilla_vit.py:111-134 and also returns softmaxed class probabilities."""
    mask_t = prepend_cls(mask_p)
    z = vit_model(x, mask_t, sd, cfg, "vit")
    phi = _vit_explainer_head(z, mask_t, sd, cfg, grand, null)
    if duo:
        return phi, softmax(linear(z[:, 0, :], sd, "classifier"))
    return phi


# ----------------------------------------------------------------------------- BERT
def bert_layer(h: np.ndarray, mask: np.ndarray, sd: SD, prefix: str, nh: int, eps: float,
               norm1_identity: bool = False) -> np.ndarray:
    """reference models/vanilla_bert.py:410-427, :556-560, :600-604 (post-LN block)."""
    ctx = self_attention(h, mask, sd, prefix + ".attention.self", nh, "bert")
    a = linear(ctx, sd, prefix + ".attention.output.dense") + h
    if not norm1_identity:
        a = layer_norm(a, sd, prefix + ".attention.output.LayerNorm", eps)
    inter = gelu(linear(a, sd, prefix + ".intermediate.dense"))
    o = linear(inter, sd, prefix + ".output.dense") + a
    return layer_norm(o, sd, prefix + ".output.LayerNorm", eps)


def bert_embeddings(ids: np.ndarray, sd: SD, prefix: str, eps: float) -> np.ndarray:
    """reference models/vanilla_bert.py:307-325 with token_type_ids == 0
    (recipes/vanilla_bert.py:289)."""
    t = ids.shape[1]
    e = sd[prefix + ".word_embeddings.weight"][ids] + sd[prefix + ".token_type_embeddings.weight"][0]
    e = e + sd[prefix + ".position_embeddings.weight"][:t][None]
    return layer_norm(e.astype(np.float32), sd, prefix + ".LayerNorm", eps)


def bert_model(ids: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict, prefix: str = "bert",
               collect: Optional[list] = None) -> np.ndarray:
    h = bert_embeddings(ids, sd, prefix + ".embeddings", cfg["layer_norm_eps"])
    if collect is not None:
        collect.append(h)
    for i in range(cfg["num_hidden_layers"]):
        h = bert_layer(h, mask_t, sd, f"{prefix}.encoder.layers.{i}", cfg["num_attention_heads"], cfg["layer_norm_eps"])
        if collect is not None:
            collect.append(h)
    return h


def bert_pool_classify(z: np.ndarray, sd: SD, pooler: str = "bert_pooler", classifier: str = "classifier",
                       act: bool = True) -> np.ndarray:
    """models/vanilla_bert.py:73-76, :615-619; act=False is the duo-BERT raw-logit head
    (models/duo_vanilla_bert.py:142-144)."""
    pooled = np.tanh(linear(z[:, 0, :], sd, pooler + ".dense"))
    logits = linear(pooled, sd, classifier)
    return softmax(logits) if act else logits


def bert_surrogate(ids: np.ndarray, mask_p: np.ndarray, sd: SD, cfg: dict, collect: Optional[list] = None) -> np.ndarray:
    z = bert_model(ids, prepend_cls(mask_p), sd, cfg, "bert", collect)
    return bert_pool_classify(z, sd)


def bert_explainer(ids: np.ndarray, mask_p: np.ndarray, grand: np.ndarray, null: np.ndarray, sd: SD, cfg: dict,
                   duo: bool = False):
    """models/vanilla_bert.py:123-162; duo: models/duo_vanilla_bert.py:117-161 (returns
    (phi, raw logits) in the recipe's swapped order, recipes/duo_vanilla_bert.py:212-213)."""
    mask_t = prepend_cls(mask_p)
    z = bert_model(ids, mask_t, sd, cfg, "bert")
    o = z
    for j in range(cfg["explainer_attn_num_layers"]):
        o = bert_layer(o, mask_t, sd, f"explainer_attn.{j}", cfg["num_attention_heads"], cfg["layer_norm_eps"],
                       norm1_identity=(j == 0))
    o = gelu(linear(o, sd, "explainer_mlp.0"))
    o = gelu(linear(o, sd, "explainer_mlp.2"))
    o = linear(o, sd, "explainer_mlp.4")
    if cfg["explainer_normalize"]:
        o = normalize_shapley_explanation(o, grand, null)
    phi = np.ascontiguousarray(o[:, 1:, :].transpose(0, 2, 1))
    if duo:
        return phi, bert_pool_classify(z, sd, act=False)
    return phi
