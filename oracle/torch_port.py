"""Oracle / CPU baseline (test infrastructure): the masked surrogate forward restated with torch *CPU* ops
(fp32, multi-threaded) — the same arithmetic library the reference itself runs on, so this is the fair
"reference CPU path" stand-in that bench.py times on the GPU box's host cores (the reference's Python cannot
travel there).  Checked against the reference-generated fixtures in tests/test_oracle_models.py.
Follows reference models/vanilla_vit.py:207-214,:242-253,:364-377,:436-465,:51-56 and
models/vanilla_bert.py:307-325,:410-427,:503-537,:61-77.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _lin(x, sd: SD, p: str):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(x, sd: SD, p: str, eps: float):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _attention(u, mask_t, sd: SD, p: str, nh: int, mode: str):
    r, t, h = u.shape
    d = h // nh
    q = _lin(u, sd, p + ".query").view(r, t, nh, d).permute(0, 2, 1, 3)
    k = _lin(u, sd, p + ".key").view(r, t, nh, d).permute(0, 2, 1, 3)
    v = _lin(u, sd, p + ".value").view(r, t, nh, d).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
    m = mask_t.to(torch.float32).reshape(r, 1, 1, t)
    s = s * m if mode == "vit" else s + (1.0 - m) * torch.finfo(torch.float32).min
    ctx = torch.matmul(F.softmax(s, dim=-1), v)
    return ctx.permute(0, 2, 1, 3).reshape(r, t, h)


def _prepend_cls(mask_p):
    return torch.cat([torch.ones((mask_p.shape[0], 1), dtype=mask_p.dtype), mask_p], dim=1)


def _vit_layer(h, mask_t, sd: SD, p: str, nh: int, eps: float, norm1_identity: bool = False):
    u = h if norm1_identity else _ln(h, sd, p + ".layernorm_before", eps)
    a = _attention(u, mask_t, sd, p + ".attention.self", nh, "vit")
    h = h + _lin(a, sd, p + ".attention.output.dense")
    w = _ln(h, sd, p + ".layernorm_after", eps)
    return _lin(F.gelu(_lin(w, sd, p + ".intermediate.dense")), sd, p + ".output.dense") + h


def _bert_layer(h, mask_t, sd: SD, p: str, nh: int, eps: float, norm1_identity: bool = False):
    ctx = _attention(h, mask_t, sd, p + ".attention.self", nh, "bert")
    a = _lin(ctx, sd, p + ".attention.output.dense") + h
    if not norm1_identity:
        a = _ln(a, sd, p + ".attention.output.LayerNorm", eps)
    return _ln(_lin(F.gelu(_lin(a, sd, p + ".intermediate.dense")), sd, p + ".output.dense") + a, sd, p + ".output.LayerNorm", eps)


def vit_backbone(x, mask_t, sd: SD, cfg: dict):
    pr = "vit.embeddings"
    e = F.conv2d(x, sd[pr + ".patch_embeddings.projection.weight"], sd[pr + ".patch_embeddings.projection.bias"],
                 stride=cfg["img_patch_size"]).flatten(2).transpose(1, 2)
    h = torch.cat([sd[pr + ".cls_token"].expand(x.shape[0], -1, -1), e], dim=1) + sd[pr + ".position_embeddings"]
    for i in range(cfg["num_hidden_layers"]):
        h = _vit_layer(h, mask_t, sd, f"vit.encoder.layers.{i}", cfg["num_attention_heads"], cfg["layer_norm_eps"])
    return _ln(h, sd, "vit.layernorm", cfg["layer_norm_eps"])


def bert_backbone(ids, mask_t, sd: SD, cfg: dict):
    t = ids.shape[1]
    pr = "bert.embeddings"
    # padding_idx: the reference's nn.Embedding(vocab, H, padding_idx=pad_token_id) (models/vanilla_bert.py:288-290) — same
    # forward, but the [PAD] row never receives a gradient
    e = F.embedding(ids, sd[pr + ".word_embeddings.weight"], padding_idx=cfg.get("pad_token_id")) + sd[pr + ".token_type_embeddings.weight"][0]
    e = e + sd[pr + ".position_embeddings.weight"][:t][None]
    h = _ln(e, sd, pr + ".LayerNorm", cfg["layer_norm_eps"])
    for i in range(cfg["num_hidden_layers"]):
        h = _bert_layer(h, mask_t, sd, f"bert.encoder.layers.{i}", cfg["num_attention_heads"], cfg["layer_norm_eps"])
    return h


def explainer_phi(x, mask_p, grand, null, sd: SD, cfg: dict, kind: str):
    """Differentiable fw_explainer (dropout off): reference models/vanilla_vit.py:102-130 / vanilla_bert.py:123-162.
    Used by the gradient parity tests (torch autograd on the CPU as the checker)."""
    mask_t = _prepend_cls(mask_p)
    nh, eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
    if kind == "vit":
        z = vit_backbone(x, mask_t, sd, cfg)
        o = z
        for j in range(cfg["explainer_attn_num_layers"]):
            o = _vit_layer(o, mask_t, sd, f"explainer_attn.{j}", nh, eps, norm1_identity=(j == 0))
        o = F.layer_norm(o, (o.shape[-1],), sd["explainer_mlp.0.weight"], sd["explainer_mlp.0.bias"], 1e-5)
        o = _lin(F.gelu(_lin(F.gelu(_lin(o, sd, "explainer_mlp.1")), sd, "explainer_mlp.3")), sd, "explainer_mlp.5")
    else:
        z = bert_backbone(x, mask_t, sd, cfg)
        o = z
        for j in range(cfg["explainer_attn_num_layers"]):
            o = _bert_layer(o, mask_t, sd, f"explainer_attn.{j}", nh, eps, norm1_identity=(j == 0))
        o = _lin(F.gelu(_lin(F.gelu(_lin(o, sd, "explainer_mlp.0")), sd, "explainer_mlp.2")), sd, "explainer_mlp.4")
    if cfg["explainer_normalize"]:
        t = o.shape[1]
        o = o + ((grand.unsqueeze(1) - null.reshape(1, 1, -1)) - o.sum(dim=1, keepdim=True)) / t
    return o[:, 1:, :].permute(0, 2, 1), z


def shapley_loss(mask, v0, vs, phi, n_players: int):
    b, k, _ = mask.shape
    pred = v0.reshape(1, 1, -1) + mask.float() @ phi.permute(0, 2, 1)
    return n_players * F.mse_loss(pred.reshape(b * k, -1), vs, reduction="mean")


@torch.no_grad()
def vit_surrogate(x, mask_p, sd: SD, cfg: dict):
    """x [R,3,px,px] fp32, mask_p [R,P] int64 -> probabilities [R,C]."""
    mask_t = _prepend_cls(mask_p)
    pr = "vit.embeddings"
    e = F.conv2d(x, sd[pr + ".patch_embeddings.projection.weight"], sd[pr + ".patch_embeddings.projection.bias"],
                 stride=cfg["img_patch_size"]).flatten(2).transpose(1, 2)
    h = torch.cat([sd[pr + ".cls_token"].expand(x.shape[0], -1, -1), e], dim=1) + sd[pr + ".position_embeddings"]
    nh, eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
    for i in range(cfg["num_hidden_layers"]):
        p = f"vit.encoder.layers.{i}"
        a = _attention(_ln(h, sd, p + ".layernorm_before", eps), mask_t, sd, p + ".attention.self", nh, "vit")
        h = h + _lin(a, sd, p + ".attention.output.dense")
        w = _ln(h, sd, p + ".layernorm_after", eps)
        h = _lin(F.gelu(_lin(w, sd, p + ".intermediate.dense")), sd, p + ".output.dense") + h
    z = _ln(h, sd, "vit.layernorm", eps)
    return F.softmax(_lin(z[:, 0, :], sd, "classifier"), dim=-1)


@torch.no_grad()
def bert_surrogate(ids, mask_p, sd: SD, cfg: dict):
    """ids [R,T] int64, mask_p [R,T-1] int64 -> probabilities [R,C]."""
    mask_t = _prepend_cls(mask_p)
    t = ids.shape[1]
    pr = "bert.embeddings"
    # padding_idx: the reference's nn.Embedding(vocab, H, padding_idx=pad_token_id) (models/vanilla_bert.py:288-290) — same
    # forward, but the [PAD] row never receives a gradient
    e = F.embedding(ids, sd[pr + ".word_embeddings.weight"], padding_idx=cfg.get("pad_token_id")) + sd[pr + ".token_type_embeddings.weight"][0]
    e = e + sd[pr + ".position_embeddings.weight"][:t][None]
    nh, eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
    h = _ln(e, sd, pr + ".LayerNorm", eps)
    for i in range(cfg["num_hidden_layers"]):
        p = f"bert.encoder.layers.{i}"
        ctx = _attention(h, mask_t, sd, p + ".attention.self", nh, "bert")
        a = _ln(_lin(ctx, sd, p + ".attention.output.dense") + h, sd, p + ".attention.output.LayerNorm", eps)
        h = _ln(_lin(F.gelu(_lin(a, sd, p + ".intermediate.dense")), sd, p + ".output.dense") + a, sd, p + ".output.LayerNorm", eps)
    pooled = torch.tanh(_lin(h[:, 0], sd, "bert_pooler.dense"))
    return F.softmax(_lin(pooled, sd, "classifier"), dim=-1)


# ----------------------------------------------------------------------------- LTT (ladder side network), differentiable
def _ltt_run(x, mask_t, sd: SD, cfg: dict, kind: str, branch: int = 0):
    """reference models/ltt_vit.py:323-340,:407-440 / models/ltt_bert.py:383-401,:468-500 -> (backbone out, side out)."""
    nh, eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
    if kind == "vit":
        pr = "vit.embeddings"
        e = F.conv2d(x, sd[pr + ".patch_embeddings.projection.weight"], sd[pr + ".patch_embeddings.projection.bias"],
                     stride=cfg["img_patch_size"]).flatten(2).transpose(1, 2)
        h = torch.cat([sd[pr + ".cls_token"].expand(x.shape[0], -1, -1), e], dim=1) + sd[pr + ".position_embeddings"]
        layer, top = _vit_layer, "vit"
    else:
        t = x.shape[1]
        pr = "bert.embeddings"
        e = sd[pr + ".word_embeddings.weight"][x] + sd[pr + ".token_type_embeddings.weight"][0]
        h = _ln(e + sd[pr + ".position_embeddings.weight"][:t][None], sd, pr + ".LayerNorm", eps)
        layer, top = _bert_layer, "bert"
    side = 0.0
    for i in range(cfg["num_hidden_layers"]):
        h = layer(h, mask_t, sd, f"{top}.encoder.layers.{i}", nh, eps)
        side = side + F.gelu(_lin(h, sd, f"{top}.encoder.s_attn_maps.{branch}_{i}"))
        side = layer(side, mask_t, sd, f"{top}.encoder.s_attn_layers.{branch}_{i}", nh, eps)
    if kind == "vit":
        return _ln(h, sd, "vit.layernorm", eps), _ln(side, sd, f"vit.s_attn_layernorm.{branch}", eps)
    return h, side


def ltt_surrogate_probs(x, mask_p, sd: SD, cfg: dict, kind: str):
    """side-branch surrogate output (models/ltt_vit.py:79-94 / ltt_bert.py:98-117), dropout off."""
    _, s = _ltt_run(x, _prepend_cls(mask_p), sd, cfg, kind)
    if kind == "vit":
        return F.softmax(_lin(s[:, 0], sd, "s_attn_classifier"), dim=-1)
    return F.softmax(_lin(torch.tanh(_lin(s[:, 0], sd, "bert_s_attn_pooler.dense")), sd, "s_attn_classifier"), dim=-1)


def ltt_explainer_phi(x, mask_p, grand, null, sd: SD, cfg: dict, kind: str):
    """fw_explainer of the LTT recipes (models/ltt_vit.py:143-183 / ltt_bert.py:167-218), dropout off -> phi [B,C,P]."""
    mask_t = _prepend_cls(mask_p)
    nh, eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
    _, o = _ltt_run(x, mask_t, sd, cfg, kind)
    if kind == "vit":
        for j in range(cfg["explainer_s_attn_num_layers"]):
            o = _vit_layer(o, mask_t, sd, f"s_explainer_attn.{j}", nh, eps, norm1_identity=(j == 0))
        o = F.layer_norm(o, (o.shape[-1],), sd["s_explainer_mlp.0.weight"], sd["s_explainer_mlp.0.bias"], 1e-5)
        o = _lin(F.gelu(_lin(F.gelu(_lin(o, sd, "s_explainer_mlp.1")), sd, "s_explainer_mlp.3")), sd, "s_explainer_mlp.5")
    else:
        for j in range(cfg["explainer_s_attn_num_layers"]):
            o = _bert_layer(o, mask_t, sd, f"s_attn_attention_layers.{j}", nh, eps, norm1_identity=(j == 0))
        o = _lin(F.gelu(_lin(F.gelu(_lin(o, sd, "s_attn_explainer.0")), sd, "s_attn_explainer.2")), sd, "s_attn_explainer.4")
    if cfg["explainer_normalize"]:
        t = o.shape[1]
        o = o + ((grand.unsqueeze(1) - null.reshape(1, 1, -1)) - o.sum(dim=1, keepdim=True)) / t
    return o[:, 1:, :].permute(0, 2, 1)
