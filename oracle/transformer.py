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


# ----------------------------------------------------------------------------- LTT (ladder side network)
def _ltt_encoder(h: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict, prefix: str, branches, layer_fn,
                 collect: Optional[list] = None):
    """reference models/ltt_vit.py:407-440 / models/ltt_bert.py:468-500: after backbone layer i, every requested
    branch b does  side_b = side_b + gelu(Linear_{b,i}(hidden));  side_b = Layer_{b,i}(side_b, mask)."""
    nh, eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
    side = {b: np.float32(0.0) for b in branches}
    for i in range(cfg["num_hidden_layers"]):
        h = layer_fn(h, mask_t, sd, f"{prefix}.encoder.layers.{i}", nh, eps)
        if collect is not None:
            collect.append(h)
        for b in branches:
            s = side[b] + gelu(linear(h, sd, f"{prefix}.encoder.s_attn_maps.{b}_{i}"))
            side[b] = layer_fn(s.astype(np.float32), mask_t, sd, f"{prefix}.encoder.s_attn_layers.{b}_{i}", nh, eps)
    return h, [side[b] for b in branches]


def ltt_vit_model(x: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict, branches, collect: Optional[list] = None):
    """reference models/ltt_vit.py:323-340 -> (LN_f(hidden), [LN_b(side_b)])."""
    h = vit_embeddings(x, sd, "vit.embeddings", cfg["img_patch_size"])
    if collect is not None:
        collect.append(h)
    h, sides = _ltt_encoder(h, mask_t, sd, cfg, "vit", branches, vit_layer, collect)
    eps = cfg["layer_norm_eps"]
    return layer_norm(h, sd, "vit.layernorm", eps), [layer_norm(s, sd, f"vit.s_attn_layernorm.{b}", eps) for b, s in zip(branches, sides)]


def ltt_vit_surrogate(x: np.ndarray, mask_p: np.ndarray, sd: SD, cfg: dict, collect: Optional[list] = None):
    """models/ltt_vit.py:79-94 -> (side probabilities, backbone probabilities)."""
    z, (s,) = ltt_vit_model(x, prepend_cls(mask_p), sd, cfg, [0], collect)
    return softmax(linear(s[:, 0, :], sd, "s_attn_classifier")), softmax(linear(z[:, 0, :], sd, "classifier"))


def _ltt_vit_head(o: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict, grand, null) -> np.ndarray:
    for j in range(cfg["explainer_s_attn_num_layers"]):
        o = vit_layer(o, mask_t, sd, f"s_explainer_attn.{j}", cfg["num_attention_heads"], cfg["layer_norm_eps"], norm1_identity=(j == 0))
    o = layer_norm(o, sd, "s_explainer_mlp.0", 1e-5)  # torch default eps (models/ltt_vit.py:124)
    o = linear(gelu(linear(gelu(linear(o, sd, "s_explainer_mlp.1")), sd, "s_explainer_mlp.3")), sd, "s_explainer_mlp.5")
    if cfg["explainer_normalize"]:
        o = normalize_shapley_explanation(o, grand, null)
    return np.ascontiguousarray(o[:, 1:, :].transpose(0, 2, 1))


def ltt_vit_explainer(x, mask_p, grand, null, sd: SD, cfg: dict):
    """models/ltt_vit.py:143-183 -> (phi [B,C,P], backbone probabilities)."""
    mask_t = prepend_cls(mask_p)
    z, (e,) = ltt_vit_model(x, mask_t, sd, cfg, [0])
    return _ltt_vit_head(e, mask_t, sd, cfg, grand, null), softmax(linear(z[:, 0, :], sd, "classifier"))


def ltt_vit_final(x, sd: SD, cfg: dict):
    """models/ltt_vit.py:231-287 (explainer_normalize=True) -> (backbone probabilities, phi)."""
    mask_t = np.ones((x.shape[0], (cfg["img_px_size"] // cfg["img_patch_size"]) ** 2 + 1), dtype=np.int64)
    z, (s, e) = ltt_vit_model(x, mask_t, sd, cfg, [0, 1])
    grand = softmax(linear(s[:, 0, :], sd, "s_attn_classifier"))
    return softmax(linear(z[:, 0, :], sd, "classifier")), _ltt_vit_head(e, mask_t, sd, cfg, grand, sd["surrogate_null"])


def ltt_bert_model(ids: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict, branches, collect: Optional[list] = None):
    """models/ltt_bert.py:383-401 (no final LayerNorms)."""
    h = bert_embeddings(ids, sd, "bert.embeddings", cfg["layer_norm_eps"])
    if collect is not None:
        collect.append(h)
    return _ltt_encoder(h, mask_t, sd, cfg, "bert", branches, bert_layer, collect)


def ltt_bert_surrogate(ids, mask_p, sd: SD, cfg: dict, collect: Optional[list] = None):
    """models/ltt_bert.py:98-117 -> (side probabilities, backbone probabilities)."""
    z, (s,) = ltt_bert_model(ids, prepend_cls(mask_p), sd, cfg, [0], collect)
    return bert_pool_classify(s, sd, "bert_s_attn_pooler", "s_attn_classifier"), bert_pool_classify(z, sd)


def _ltt_bert_head(o: np.ndarray, mask_t: np.ndarray, sd: SD, cfg: dict, grand, null) -> np.ndarray:
    for j in range(cfg["explainer_s_attn_num_layers"]):
        o = bert_layer(o, mask_t, sd, f"s_attn_attention_layers.{j}", cfg["num_attention_heads"], cfg["layer_norm_eps"], norm1_identity=(j == 0))
    o = linear(gelu(linear(gelu(linear(o, sd, "s_attn_explainer.0")), sd, "s_attn_explainer.2")), sd, "s_attn_explainer.4")
    if cfg["explainer_normalize"]:
        o = normalize_shapley_explanation(o, grand, null)
    return np.ascontiguousarray(o[:, 1:, :].transpose(0, 2, 1))


def ltt_bert_explainer(ids, mask_p, grand, null, sd: SD, cfg: dict):
    """models/ltt_bert.py:167-218 -> (phi, backbone probabilities)."""
    mask_t = prepend_cls(mask_p)
    z, (e,) = ltt_bert_model(ids, mask_t, sd, cfg, [0])
    return _ltt_bert_head(e, mask_t, sd, cfg, grand, null), bert_pool_classify(z, sd)


def ltt_bert_final(ids, sd: SD, cfg: dict):
    """models/ltt_bert.py:258-338 (explainer_normalize=True) -> (backbone probabilities, phi)."""
    mask_t = np.ones_like(ids)
    z, (s, e) = ltt_bert_model(ids, mask_t, sd, cfg, [0, 1])
    grand = bert_pool_classify(s, sd, "bert_s_attn_pooler", "s_attn_classifier")
    return bert_pool_classify(z, sd), _ltt_bert_head(e, mask_t, sd, cfg, grand, sd["surrogate_null"])
