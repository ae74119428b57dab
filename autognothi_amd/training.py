"""Training forward/backward of the masked transformers on the HIP kernels (fp32).

Replaces what torch.autograd does for the reference in scripts/train_explainer.py:184-198
(explainer: backbone with all-ones mask -> explainer_attn -> MLP -> normalise -> Shapley loss) and
scripts/train_surrogate.py:145-147 (surrogate: masked forward -> KL against the frozen classifier).
Gradients are accumulated into the modules' ``param.grad`` (fp32, on the device) so the reference's
``torch.optim.AdamW`` + ``CosineAnnealingLR`` drive the update unchanged; frozen parameters
(``requires_grad == False``, e.g. the froyo backbone) are skipped, and a fully frozen backbone is run through
the fast inference path without saving activations.

All arithmetic is libautognothi_hip kernels: Linear forward/backward = ag_gemm (dX against the transposed
weight, dW against transposed activations), plus the train.hip blocks.  Activations are kept in fp32.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _lib as L
from . import engine, ops

F32 = L.AG_F32
PAD = 32  # ag_gemm fp32 needs its K dimension to be a multiple of 32

# Mixed precision for the training GEMMs (opt-in: AG_TRAIN_BF16=1 or training.MIXED_BF16 = True).  Activations, gradients,
# parameters and every elementwise / attention kernel stay fp32; only the operands of the forward, dX and dW GEMMs are
# rounded to bf16 (fp32 accumulate, fp32 output) — the semantics of torch.autocast(bf16) around nn.Linear.  The reference
# trains in fp32, so this is a throughput mode like the bf16 inference path, off by default; the gradient-parity tests
# run the exact-fp32 kernels.
import os as _os
MIXED_BF16 = _os.environ.get("AG_TRAIN_BF16", "0") == "1"


def _mm(a: Tensor, w: Tensor, bias: Optional[Tensor], epilogue: int, m: int, w_bf16: Optional[Tensor] = None) -> Tensor:
    """epilogue(a[m,K] @ w[N,K]^T + bias) -> fp32; bf16 operands when MIXED_BF16 (plain / fp32-output epilogues only).
    ``w_bf16``: an already cast copy of ``w`` (Lin caches one per parameter version)."""
    if MIXED_BF16 and epilogue in (L.AG_EPI_BIAS, L.AG_EPI_BIAS_F32) and a.shape[-1] % 8 == 0:
        wb = w_bf16 if w_bf16 is not None else ops.cast(w, L.AG_BF16)
        return ops.gemm(ops.cast(a, L.AG_BF16), wb, bias, L.AG_EPI_BIAS_F32, L.AG_BF16, m=m)
    return ops.gemm(a, w, bias, epilogue, F32, m=m)


# Called with every parameter whose gradient of the current step is final (each parameter receives exactly one contribution
# per step): distributed.GradBucketReducer.ready during N-rank training, so that a bucket's all-reduce starts while the
# backward of the layers below is still running.  None: nothing to tell.
GRAD_SINK: Optional[Callable[[Tensor], None]] = None


def _final(*params: Tensor) -> None:
    if GRAD_SINK is not None:
        for p in params:
            if p.requires_grad:
                GRAD_SINK(p)


def _touch(p: Tensor) -> None:
    """a gradient is being written into p: whatever optimiser runs next will change its value, whether or not it advances
    ``p._version`` (fused AdamW does not) — engine.param_key counts these writes so that every weight cache (operand forms
    here, the inference packs in engine.py) is rebuilt before the next forward."""
    p.__dict__["_ag_step"] = p.__dict__.get("_ag_step", 0) + 1


def _grad(p: Tensor) -> Tensor:
    _touch(p)
    if p.grad is None:
        p.grad = torch.zeros_like(p, dtype=torch.float32)
    return p.grad


def _acc_grad(p: Tensor, src: Tensor, fresh: bool = False) -> None:
    """p.grad += src.  After ``zero_grad(set_to_none=True)`` (torch's default) the first contribution of a step becomes the
    gradient buffer itself (``fresh``: src is a whole tensor nobody else holds) or one copy of it — instead of a zero fill
    plus an add per parameter and step."""
    src = src.reshape(p.shape)
    _touch(p)
    if p.grad is None:
        p.grad = src if (fresh and src.is_contiguous()) else src.clone(memory_format=torch.contiguous_format)
    else:
        _acc(p.grad, src)
    _final(p)


def _acc(dst: Tensor, src: Tensor) -> None:
    """dst += src through the add kernel (no torch arithmetic)."""
    with L.on(dst.device):
        L.check(L.lib().ag_add_f32(L.ptr(dst), L.ptr(src.contiguous()), L.ptr(dst), dst.numel(), L.stream()))


def _fp32_step(fn):
    """Training steps run the fp32 kernels; the caller's inference precision (bf16 throughput mode for the K-mask
    surrogate targets of the same batch loop) is restored afterwards."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        prev = engine.precision_name()
        try:
            return fn(*args, **kwargs)
        finally:
            engine.set_precision(prev)
    return wrapper


class Seeds:
    """Fresh dropout seed per site per step (train mode); p == 0 disables everything."""

    def __init__(self, base: int):
        self.base, self.n = base & 0x7FFFFFFF, 0

    def next(self) -> int:
        self.n += 1
        return (self.base * 2654435761 + self.n * 40503) & 0xFFFFFFFF


class Lin:
    """y = x Wᵀ + b over one or more nn.Linear modules fused along the output dim (q|k|v)."""

    def __init__(self, mods: Sequence[nn.Module]):
        self.mods = list(mods)
        self.x: Optional[Tensor] = None
        self.xt: Optional[Tensor] = None      # mixed mode: X^T bf16 [K, Mp] saved instead of x (the dW operand)

    def _key(self):
        return tuple(engine.param_key(p) for m in self.mods for p in (m.weight, m.bias))

    def _w(self) -> Tuple[Tensor, Tensor]:
        """fused fp32 [N,K] weight and [N] bias, rebuilt only when a parameter changed (optimizer step / load): the forward
        and the backward of a step, and every step of a frozen module, share one copy."""
        key = self._key()
        if getattr(self, "_wkey", None) != key:
            if len(self.mods) == 1:
                m = self.mods[0]
                self._wb = (m.weight.detach().reshape(m.weight.shape[0], -1).float().contiguous(), m.bias.detach().float().contiguous())
            else:
                self._wb = (torch.cat([m.weight.detach() for m in self.mods], 0).float().contiguous(),
                            torch.cat([m.bias.detach() for m in self.mods], 0).float().contiguous())
            self._wkey, self._derived = key, {}
        return self._wb

    def _w_form(self, form: str) -> Optional[Tensor]:
        """derived operand forms of the current weight, cached with it: "bf16" [N,K], "t" fp32 [K,Np], "t_bf16"."""
        w, _ = self._w()
        d = self._derived
        if form not in d:
            if form == "bf16":
                d[form] = ops.cast(w, L.AG_BF16)
            elif form == "t":
                d[form] = ops.transpose(w, pad_cols_to=PAD) if w.shape[0] % PAD else ops.transpose(w)
            elif form == "t_bf16":     # transpose + round in one pass (no fp32 transposed copy, no cast launch)
                d[form] = ops.transpose_bf16(w, pad_cols_to=PAD)
        return d[form]

    def trainable(self) -> bool:
        return any(p.requires_grad for m in self.mods for p in (m.weight, m.bias))

    def _w_pair(self) -> Tuple[Tensor, Tensor]:
        """(W bf16 [N,K], W^T bf16 [K,Np]) of the current weight from ONE launch (they change with every optimiser step)."""
        w, _ = self._w()
        d = self._derived
        if "pair" not in d:
            d["pair"] = ops.cast_transpose_bf16(w, pad_cols_to=PAD)
        return d["pair"]

    def forward(self, x: Tensor, epilogue: int = L.AG_EPI_BIAS, save: bool = True) -> Tensor:
        w, b = self._w()
        mixed = MIXED_BF16 and epilogue in (L.AG_EPI_BIAS, L.AG_EPI_BIAS_F32) and x.shape[-1] % 8 == 0
        self.xt = None
        if mixed:
            # bf16 operands, fp32 accumulate / output.  A trainable Linear needs x^T for dW later: both forms in one launch
            wb = self._w_pair()[0]
            if save and self.trainable():
                xb, self.xt = ops.cast_transpose_bf16(x, pad_cols_to=PAD)
            else:
                xb = ops.cast(x, L.AG_BF16)
                if save:
                    self.x = x
            return ops.gemm(xb, wb, b, L.AG_EPI_BIAS_F32, L.AG_BF16, m=x.shape[0])
        if save:
            self.x = x
        return _mm(x, w, b, epilogue, x.shape[0], None)

    def forward_gelu(self, u: Tensor) -> Tensor:
        """forward(gelu(u)).  Mixed mode, trainable: GELU is applied inside the pass that makes the two bf16 operand forms, the
        fp32 GELU output never exists (the backward recomputes gelu' from u: Block keeps u)."""
        if MIXED_BF16 and self.trainable() and u.shape[-1] % 8 == 0:
            _, b = self._w()
            xb, self.xt = ops.gelu_cast_transpose_bf16(u, pad_cols_to=PAD)
            return ops.gemm(xb, self._w_pair()[0], b, L.AG_EPI_BIAS_F32, L.AG_BF16, m=u.shape[0])
        return self.forward(ops.gelu(u))

    def backward_gelu(self, u: Tensor, dg: Tensor, need_dx: bool = True) -> Optional[Tensor]:
        """backward(dg * gelu'(u)) for this Linear's OUTPUT pre-activation u (fc1): mixed mode fuses gelu' into the operand pass."""
        w, _ = self._w()
        n, k = w.shape
        if MIXED_BF16 and self.trainable() and n % PAD == 0 and k % 8 == 0:
            du, dyb, dyt = ops.gelu_bwd_cast_transpose_bf16(u, dg, pad_cols_to=PAD)
            return self.backward(du, need_dx, pre=(dyb, dyt))
        return self.backward(ops.gelu_bwd(u, dg), need_dx)

    def backward(self, dy: Tensor, need_dx: bool = True, pre: Optional[Tuple[Tensor, Tensor]] = None) -> Optional[Tensor]:
        w, _ = self._w()
        n, k = w.shape
        m = dy.shape[0]
        dx = None
        train = self.trainable()
        if MIXED_BF16 and n % PAD == 0 and k % 8 == 0:
            # dy and dy^T as bf16 from one launch; dX = dY . W against W^T [K, N], dW = dY^T . X against X^T [K, Mp]
            if pre is not None:
                dyb, dyt = pre
            elif train:
                dyb, dyt = ops.cast_transpose_bf16(dy, pad_cols_to=PAD)
            else:
                dyb, dyt = ops.cast(dy, L.AG_BF16), None
            if need_dx:
                dx = ops.gemm(dyb, self._w_pair()[1], None, L.AG_EPI_BIAS_F32, L.AG_BF16, m=m)
            if train:
                xt = self.xt if self.xt is not None else ops.transpose_bf16(self.x, pad_cols_to=PAD)
                dw = ops.gemm(dyt, xt, None, L.AG_EPI_BIAS_F32, L.AG_BF16, m=n)
        else:
            if need_dx:
                # dX[M,K] = dY[M,N] · W[N,K]  ==  NT GEMM against Wᵀ [K, N]; pad N (the contraction) to 32
                wt = self._w_form("t")
                wt_b = self._w_form("t_bf16") if MIXED_BF16 else None
                if n % PAD:
                    npad = (n + PAD - 1) // PAD * PAD
                    dyp = torch.zeros((m, npad), dtype=torch.float32, device=dy.device)
                    dyp[:, :n].copy_(dy)
                    dx = _mm(dyp, wt, None, L.AG_EPI_BIAS, m, wt_b)
                else:
                    dx = _mm(dy, wt, None, L.AG_EPI_BIAS, m, wt_b)
            if train:
                # dW[N,K] = dYᵀ[N,M] · X[M,K]  ==  NT GEMM of dYᵀ [N,Mp] against Xᵀ [K,Mp]
                if MIXED_BF16:   # (Mp % 32 == 0 keeps the bf16 rows 16-byte aligned)
                    xt = self.xt if self.xt is not None else ops.transpose_bf16(self.x, pad_cols_to=PAD)
                    dw = ops.gemm(ops.transpose_bf16(dy, pad_cols_to=PAD), xt, None, L.AG_EPI_BIAS_F32, L.AG_BF16, m=n)
                else:
                    dw = _mm(ops.transpose(dy, pad_cols_to=PAD), ops.transpose(self.x, pad_cols_to=PAD), None, L.AG_EPI_BIAS, n)
        if train:
            db = ops.colsum(dy)
            off = 0
            for mod in self.mods:
                rows = mod.weight.shape[0]
                whole = len(self.mods) == 1
                # (row slices of the fused q|k|v gradient are contiguous and nobody else holds dw / db: they become the .grad
                # tensors themselves, three views of one buffer, instead of three copies)
                if mod.weight.requires_grad:
                    _acc_grad(mod.weight, dw[off:off + rows] if not whole else dw, fresh=True)
                if mod.bias.requires_grad:
                    _acc_grad(mod.bias, db[off:off + rows] if not whole else db, fresh=True)
                off += rows
        self.x = None
        self.xt = None
        return dx


class Norm:
    def __init__(self, mod: nn.Module, eps: float):
        self.mod, self.eps = mod, eps
        self.identity = isinstance(mod, nn.Identity)
        self.x: Optional[Tensor] = None

    def forward(self, x: Tensor) -> Tensor:
        if self.identity:
            return x
        self.x = x
        _, y = ops.layernorm(x, self.mod.weight.detach().float(), self.mod.bias.detach().float(), self.eps, F32,
                             want_store=False, want_f32=True)
        return y

    def backward(self, dy: Tensor, add: Optional[Tensor] = None) -> Tensor:
        """-> dx (+ ``add``: a gradient that joins over the residual branch, summed in the same pass)."""
        if self.identity:
            return dy if add is None else ops.add(dy, add)
        train = self.mod.weight.requires_grad
        fresh = train and self.mod.weight.grad is None and self.mod.bias.grad is None
        if fresh:   # first contribution of the step: the kernel stores instead of accumulating (no zero fill)
            self.mod.weight.grad = torch.empty_like(self.mod.weight, dtype=torch.float32)
            self.mod.bias.grad = torch.empty_like(self.mod.bias, dtype=torch.float32)
        dx = ops.layernorm_bwd(self.x, self.mod.weight.detach().float(), dy, self.eps,
                               _grad(self.mod.weight) if train else None, _grad(self.mod.bias) if train else None, accumulate=not fresh,
                               add=add)
        self.x = None
        if train:
            _final(self.mod.weight, self.mod.bias)
        return dx


class Block:
    """One transformer layer (reference VanillaViTLayer :364-377 pre-LN / VanillaBertLayer :410-427 post-LN)."""

    def __init__(self, layer: nn.Module, kind: int, heads: int, eps: float, p_hidden: float, p_attn: float):
        self.kind, self.heads, self.p_hidden, self.p_attn = kind, heads, p_hidden, p_attn
        att = layer.attention
        self.qkv = Lin([att.self.query, att.self.key, att.self.value])
        self.o, self.fc1, self.fc2 = Lin([att.output.dense]), Lin([layer.intermediate.dense]), Lin([layer.output.dense])
        if kind == L.AG_MASK_VIT_MUL:
            self.n1, self.n2 = Norm(layer.layernorm_before, eps), Norm(layer.layernorm_after, eps)
        else:
            self.n1, self.n2 = Norm(att.output.LayerNorm, eps), Norm(layer.output.LayerNorm, eps)
        self.saved = None

    def forward(self, h: Tensor, bits: Tensor, rows: int, t: int, seeds: Seeds, train: bool) -> Tensor:
        hdim = h.shape[-1]
        ph, pa = (self.p_hidden, self.p_attn) if train else (0.0, 0.0)
        s_att, s_o, s_f = seeds.next(), seeds.next(), seeds.next()
        vit = self.kind == L.AG_MASK_VIT_MUL
        u1 = self.n1.forward(h) if vit else h
        qkv = self.qkv.forward(u1)
        ctx = ops.masked_attention_train(qkv, bits, rows, t, hdim, self.heads, self.kind, pa, s_att, mixed=MIXED_BF16).view(rows * t, hdim)
        hx = ops.dropout_add(self.o.forward(ctx), h, ph, s_o)         # h + dropout(dense(ctx)) in one pass
        u2 = self.n2.forward(hx) if vit else self.n1.forward(hx)
        f1 = self.fc1.forward(u2)
        out = ops.dropout_add(self.fc2.forward_gelu(f1), hx if vit else u2, ph, s_f)
        if not vit:
            out = self.n2.forward(out)
        self.saved = (qkv, ctx, f1, bits, rows, t, hdim, ph, pa, s_att, s_o, s_f)
        return out

    def backward(self, dout: Tensor) -> Tensor:
        qkv, ctx, f1, bits, rows, t, hdim, ph, pa, s_att, s_o, s_f = self.saved
        vit = self.kind == L.AG_MASK_VIT_MUL
        if vit:
            du2 = self.fc1.backward_gelu(f1, self.fc2.backward(ops.dropout(dout, ph, s_f)))
            dhx = self.n2.backward(du2, add=dout)              # + the gradient over the residual branch, in the same pass
            dctx = self.o.backward(ops.dropout(dhx, ph, s_o))
            dqkv = ops.masked_attention_bwd(qkv.view(rows, t, 3 * hdim), bits, ctx.view(rows, t, hdim), dctx.view(rows, t, hdim),
                                            rows, t, hdim, self.heads, self.kind, pa, s_att, mixed=MIXED_BF16).view(rows * t, 3 * hdim)
            du1 = self.qkv.backward(dqkv)
            dh = self.n1.backward(du1, add=dhx)
        else:
            dsum = self.n2.backward(dout)                       # d(f2 + a)
            da = self.fc1.backward_gelu(f1, self.fc2.backward(ops.dropout(dsum, ph, s_f)))
            da = ops.add(da, dsum)
            dpre = self.n1.backward(da)                         # d(ao + h)
            dctx = self.o.backward(ops.dropout(dpre, ph, s_o))
            dqkv = ops.masked_attention_bwd(qkv.view(rows, t, 3 * hdim), bits, ctx.view(rows, t, hdim), dctx.view(rows, t, hdim),
                                            rows, t, hdim, self.heads, self.kind, pa, s_att, mixed=MIXED_BF16).view(rows * t, 3 * hdim)
            dh = ops.add(dpre, self.qkv.backward(dqkv))
        self.saved = None
        return dh


def _any_trainable(mod: nn.Module) -> bool:
    return any(p.requires_grad for p in mod.parameters())


def _module_n_players(m: nn.Module) -> int:
    """recipe.n_players(cfg) read off the module: ViT = patches (recipes/vanilla_vit.py n_players), BERT =
    max_position_embeddings - 1 (recipes/vanilla_bert.py)."""
    cfg = m.config
    if hasattr(cfg, "img_px_size"):
        return (cfg.img_px_size // cfg.img_patch_size) ** 2
    return cfg.max_position_embeddings - 1


def _drop_block_saved(blk: "Block") -> None:
    blk.saved = None
    for lin in (blk.qkv, blk.o, blk.fc1, blk.fc2):
        lin.x = lin.xt = None
    blk.n1.x = blk.n2.x = None


class ViTBackboneTrainer:
    """VanillaViTModel (embeddings + encoder + final LN), reference models/vanilla_vit.py:207-214."""

    def __init__(self, vit: nn.Module):
        self.vit, c = vit, vit.config
        self.frozen = not _any_trainable(vit)
        self.blocks = [Block(ly, L.AG_MASK_VIT_MUL, c.num_attention_heads, c.layer_norm_eps, c.hidden_dropout_prob,
                             c.attention_probs_dropout_prob) for ly in vit.encoder.layers]
        self.ln_f = Norm(vit.layernorm, c.layer_norm_eps)
        self.proj = Lin([vit.embeddings.patch_embeddings.projection])
        self.saved = None

    def forward(self, x: Tensor, bits: Tensor, seeds: Seeds, train: bool, tap: Optional[Callable[[int, Tensor], None]] = None) -> Tensor:
        """-> LN_final(hidden) fp32 [B*T, H].  ``tap(i, hidden_i)`` is called after layer i (LTT ladder)."""
        c = self.vit.config
        b, p, h = x.shape[0], self.vit.n_players, c.hidden_size
        t = p + 1
        if self.frozen:
            # eval()-equivalent: a frozen backbone has no dropout-free requirement in the reference, but its
            # activations need no saving; keep dropout semantics by using the training path only when p > 0
            pass
        x = x.contiguous().float()
        cols = torch.empty((b * p, c.img_channels * c.img_patch_size ** 2), dtype=torch.float32, device=x.device)
        with L.on(x.device):
            L.check(L.lib().ag_vit_im2col(L.ptr(x), b, c.img_channels, c.img_px_size, c.img_patch_size, L.ptr(cols), F32, L.stream()))
            pe = self.proj.forward(cols, L.AG_EPI_BIAS_F32)
            h0 = torch.empty((b, t, h), dtype=torch.float32, device=x.device)
            e = self.vit.embeddings
            L.check(L.lib().ag_vit_assemble(L.ptr(pe), L.ptr(e.cls_token.detach().float().contiguous()),
                                            L.ptr(e.position_embeddings.detach().float().contiguous()), b, p, h, L.ptr(h0), L.stream()))
        s_emb = seeds.next()
        ph = c.hidden_dropout_prob if train else 0.0
        hid = ops.dropout(h0.view(b * t, h), ph, s_emb)
        for i, blk in enumerate(self.blocks):
            hid = blk.forward(hid, bits, b, t, seeds, train)
            if tap is not None:
                tap(i, hid)
        self.saved = (b, p, h, ph, s_emb)
        return self.ln_f.forward(hid)

    def drop_saved(self) -> None:
        """forget the activations of a forward whose backward will not run (frozen backbone)."""
        for blk in self.blocks:
            _drop_block_saved(blk)
        self.ln_f.x = self.proj.x = self.proj.xt = None
        self.saved = None

    def backward(self, dz: Tensor) -> None:
        b, p, h, ph, s_emb = self.saved
        t = p + 1
        d = self.ln_f.backward(dz)
        for blk in reversed(self.blocks):
            d = blk.backward(d)
        d = ops.dropout(d, ph, s_emb).view(b, t, h)
        e = self.vit.embeddings
        if e.position_embeddings.requires_grad:
            _acc_grad(e.position_embeddings, ops.colsum(d.reshape(b, t * h)), fresh=True)
        if e.cls_token.requires_grad:
            _acc_grad(e.cls_token, ops.colsum(d[:, 0, :].contiguous()), fresh=True)
        if self.proj.trainable():
            self.proj.backward(d[:, 1:, :].contiguous().view(b * p, h), need_dx=False)
        self.saved = None


class BertBackboneTrainer:
    """VanillaBertModel (embeddings + LN + encoder), reference models/vanilla_bert.py:248-268."""

    def __init__(self, bert: nn.Module):
        self.bert, c = bert, bert.config
        self.blocks = [Block(ly, L.AG_MASK_BERT_ADD, c.num_attention_heads, c.layer_norm_eps, c.hidden_dropout_prob,
                             c.attention_probs_dropout_prob) for ly in bert.encoder.layers]
        self.ln_e = Norm(bert.embeddings.LayerNorm, c.layer_norm_eps)
        self.saved = None

    def forward(self, ids: Tensor, bits: Tensor, seeds: Seeds, train: bool, tap: Optional[Callable[[int, Tensor], None]] = None) -> Tensor:
        c, e = self.bert.config, self.bert.embeddings
        ids = ids.contiguous().to(torch.int64)
        b, t = ids.shape
        h = c.hidden_size
        # gather (index plumbing) + add through the add kernel, then LayerNorm
        emb = e.word_embeddings.weight.detach().float()[ids].view(b * t, h).contiguous()
        pos_type = ops.add(e.position_embeddings.weight.detach().float()[:t].contiguous(),
                           e.token_type_embeddings.weight.detach().float()[0:1].expand(t, h).contiguous())
        emb = ops.add(emb, pos_type.repeat(b, 1).contiguous())
        s_emb = seeds.next()
        ph = c.hidden_dropout_prob if train else 0.0
        hid = ops.dropout(self.ln_e.forward(emb), ph, s_emb)
        for i, blk in enumerate(self.blocks):
            hid = blk.forward(hid, bits, b, t, seeds, train)
            if tap is not None:
                tap(i, hid)
        self.saved = (ids, b, t, h, ph, s_emb)
        return hid

    def drop_saved(self) -> None:
        for blk in self.blocks:
            _drop_block_saved(blk)
        self.ln_e.x = None
        self.saved = None

    def backward(self, dh: Tensor) -> None:
        ids, b, t, h, ph, s_emb = self.saved
        d = dh
        for blk in reversed(self.blocks):
            d = blk.backward(d)
        d = self.ln_e.backward(ops.dropout(d, ph, s_emb))
        e = self.bert.embeddings
        if e.position_embeddings.weight.requires_grad:
            g = _grad(e.position_embeddings.weight)
            _acc(g[:t], ops.colsum(d.view(b, t * h)).view(t, h))
            _final(e.position_embeddings.weight)
        if e.token_type_embeddings.weight.requires_grad:
            g = _grad(e.token_type_embeddings.weight)
            _acc(g[0], ops.colsum(d))
            _final(e.token_type_embeddings.weight)
        if e.word_embeddings.weight.requires_grad:
            # scatter-add of B*T rows into the vocabulary table: index plumbing (torch), no arithmetic beyond the adds.
            # nn.Embedding(padding_idx=pad_token_id) (reference models/vanilla_bert.py:288-290): autograd never gives the
            # [PAD] row a gradient, although [PAD] positions are ordinary players in the forward.
            pad = e.word_embeddings.padding_idx
            flat = ids.view(-1)
            if pad is not None:
                d = d.masked_fill((flat == pad).unsqueeze(1), 0.0)
            _grad(e.word_embeddings.weight).index_add_(0, flat, d)
            _final(e.word_embeddings.weight)
        self.saved = None


class MLPHead:
    """explainer_mlp: ViT [LN, Linear, GELU, Linear, GELU, Linear] (:92-100) / BERT without the LN (:114-121)."""

    def __init__(self, seq: nn.Sequential):
        mods = list(seq)
        self.ln = Norm(mods[0], mods[0].eps) if isinstance(mods[0], nn.LayerNorm) else None
        lins = [m for m in mods if isinstance(m, nn.Linear)]
        self.l1, self.l2, self.l3 = Lin([lins[0]]), Lin([lins[1]]), Lin([lins[2]])
        self.saved = None

    def forward(self, x: Tensor) -> Tensor:
        if self.ln is not None:
            x = self.ln.forward(x)
        a = self.l1.forward(x)
        ga = ops.gelu(a)
        bb = self.l2.forward(ga)
        gb = ops.gelu(bb)
        self.saved = (a, bb)
        return self.l3.forward(gb, L.AG_EPI_BIAS_F32)

    def backward(self, dpred: Tensor) -> Tensor:
        a, bb = self.saved
        d = self.l3.backward(dpred)
        d = self.l2.backward(ops.gelu_bwd(bb, d))
        d = self.l1.backward(ops.gelu_bwd(a, d))
        if self.ln is not None:
            d = self.ln.backward(d)
        self.saved = None
        return d


def _cross_entropy(logits: Tensor, labels: Tensor) -> Tuple[Tensor, Tensor]:
    """F.cross_entropy(x, labels) (mean) and its gradient wrt x, from the soft-max kernel; the one-hot
    subtraction is index plumbing on a [B,C] tensor."""
    s = ops.softmax_rows(logits)
    b = logits.shape[0]
    picked = s.gather(1, labels.view(-1, 1).to(torch.int64))
    loss = -(picked.log()).mean()
    g = s.clone()
    g.scatter_add_(1, labels.view(-1, 1).to(torch.int64), torch.full((b, 1), -1.0, device=s.device))
    return loss, g / b


def _bf16_step(module: nn.Module) -> bool:
    """MIXED_BF16 and a model the bf16-activation step of training16.py covers (vanilla / duo / froyo, head dim 64)."""
    if not MIXED_BF16 or type(module).__name__.startswith("Ltt"):
        return False
    from . import training16
    return training16.supported(module)


class ExplainerTrainer:
    """fw_explainer + loss_shapley_new with gradients (vanilla / froyo / duo; ViT or BERT).  With MIXED_BF16 on, constructing
    one returns the bf16-activation trainer of training16.py (same interface) for the models it covers."""

    def __new__(cls, recipe, m_explainer: nn.Module):
        if cls is ExplainerTrainer and _bf16_step(m_explainer):
            from . import training16
            return training16.ExplainerTrainer16(recipe, m_explainer)
        return super().__new__(cls)

    def __init__(self, recipe, m_explainer: nn.Module):
        """``recipe`` may be None (the autograd bridge builds trainers from the module alone)."""
        self.recipe, self.m = recipe, m_explainer
        cfg = m_explainer.config
        self.is_vit = hasattr(m_explainer, "vit")
        self.duo = bool(recipe.training.exp_variant_duo) if recipe is not None else hasattr(m_explainer, "classifier")
        self.n_players = _module_n_players(m_explainer)
        self.saved = None
        self.kind = L.AG_MASK_VIT_MUL if self.is_vit else L.AG_MASK_BERT_ADD
        self.backbone = ViTBackboneTrainer(m_explainer.vit) if self.is_vit else BertBackboneTrainer(m_explainer.bert)
        self.backbone_frozen = not _any_trainable(m_explainer.vit if self.is_vit else m_explainer.bert)
        self.attn = [Block(ly, self.kind, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.hidden_dropout_prob,
                           cfg.attention_probs_dropout_prob) for ly in m_explainer.explainer_attn]
        self.mlp = MLPHead(m_explainer.explainer_mlp)
        self.cls = Lin([m_explainer.classifier]) if self.duo else None
        self.pool = Lin([m_explainer.bert_pooler.dense]) if (self.duo and not self.is_vit) else None
        self.step = 0

    def forward_phi(self, xs: Tensor, v_0: Optional[Tensor], v_1: Optional[Tensor], train: bool = True, seed: int = 0,
                    bits: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """fw_explainer with activations saved for ``backward_phi`` -> (phi [B,C,P], base_Ys or None).  ``base_Ys`` is what the
        duo recipes return second: soft-maxed head output for ViT (models/duo_vanilla_vit.py:121-122), raw logits for BERT
        (models/duo_vanilla_bert.py:142-144).  ``bits``: key bits of the attention mask (all ones in every reference caller)."""
        cfg = self.m.config
        self.step += 1
        seeds = Seeds(seed * 7919 + self.step)
        b = xs.shape[0]
        p = self.n_players
        t, h, c = p + 1, cfg.hidden_size, cfg.num_labels
        if bits is None:
            bits = engine.ones_mask_bits(b, p, xs.device)
        z = self.backbone.forward(xs, bits, seeds, train)          # [B*T, H] (ViT: after the final LN)
        o = z
        for blk in self.attn:
            o = blk.forward(o, bits, b, t, seeds, train)
        s_exp = seeds.next()
        ph = cfg.hidden_dropout_prob if (train and not self.is_vit) else 0.0   # BERT explainer_dropout (:152)
        o = ops.dropout(o, ph, s_exp)
        pred = self.mlp.forward(o).view(b, t, c)
        phi = ops.shapley_normalize(pred, v_1, v_0, normalize=bool(cfg.explainer_normalize))
        base, duo_saved = None, None
        if self.duo:
            zc = z.view(b, t, h)[:, 0, :].contiguous()
            if self.is_vit:
                base = ops.softmax_rows(self.cls.forward(zc, L.AG_EPI_BIAS_F32))
                duo_saved = (base,)
            else:
                pooled = self.pool.forward(zc, L.AG_EPI_BIAS_TANH)
                s_pool = seeds.next()
                pd = cfg.hidden_dropout_prob if train else 0.0
                base = self.cls.forward(ops.dropout(pooled, pd, s_pool), L.AG_EPI_BIAS_F32)
                duo_saved = (pooled, pd, s_pool)
        self.saved = (b, t, h, c, ph, s_exp, duo_saved)
        return phi, base

    def backward_phi(self, dphi: Tensor, dbase: Optional[Tensor] = None) -> None:
        """Backward of ``forward_phi`` into param.grad: dphi = d loss / d phi [B,C,P]; dbase = d loss / d base_Ys (duo)."""
        cfg = self.m.config
        b, t, h, c, ph, s_exp, duo_saved = self.saved
        self.saved = None
        dz_extra = None
        if self.duo and dbase is not None:
            if self.is_vit:
                (probs,) = duo_saved
                dz_cls = self.cls.backward(ops.softmax_rows_bwd(probs, dbase.contiguous().float()))
            else:
                pooled, pd, s_pool = duo_saved
                dp = ops.dropout(self.cls.backward(dbase.contiguous().float()), pd, s_pool)
                dz_cls = self.pool.backward(ops.tanh_bwd(pooled, dp))
            dz_extra = torch.zeros((b, t, h), dtype=torch.float32, device=dphi.device)
            dz_extra[:, 0, :].copy_(dz_cls)
        elif self.duo:
            self.cls.x = self.cls.xt = None
            if self.pool is not None:
                self.pool.x = self.pool.xt = None
        dpred = ops.shapley_normalize_bwd(dphi, t, normalize=bool(cfg.explainer_normalize)).view(b * t, c)
        d = ops.dropout(self.mlp.backward(dpred), ph, s_exp)
        for blk in reversed(self.attn):
            d = blk.backward(d)
        if dz_extra is not None:
            d = ops.add(d, dz_extra.view(b * t, h))
        if not self.backbone_frozen:
            self.backbone.backward(d)
        else:
            self.backbone.drop_saved()

    @_fp32_step
    def loss_and_grads(self, xs: Tensor, bits_loss: Tensor, v_0: Tensor, v_s: Tensor, v_1: Tensor, n_mask_samples: int,
                       labels: Optional[Tensor] = None, train: bool = True, seed: int = 0):
        """One reference training-step body (scripts/train_explainer.py:182-196 / train_duo_explainer.py:180-196):
        explainer forward (all-ones mask), loss, backward into param.grad.  -> (loss tensor [1], phi)."""
        b = xs.shape[0]
        phi, base = self.forward_phi(xs, v_0, v_1, train, seed)
        loss, dphi = ops.shapley_loss(bits_loss, v_0, v_s, phi, b, n_mask_samples, want_grad=True)
        total, dbase = loss, None
        self.last_parts = (loss, None, base)               # (Shapley loss, classification loss, base_Ys) of this step: duo reports
        if self.duo:
            ce, dbase = _cross_entropy(base, labels)     # duo-ViT: CE on probabilities, duo-BERT: on raw logits (A.5)
            total = loss + ce
            self.last_parts = (loss, ce, base)
        self.backward_phi(dphi, dbase)
        return total, phi


class SurrogateTrainer:
    """fw_surrogate on masked inputs + loss_logits_kl_divergence with gradients (scripts/train_surrogate.py:133-147).  With
    MIXED_BF16 on: the bf16-activation trainer of training16.py for the models it covers."""

    def __new__(cls, recipe, m_surrogate: nn.Module):
        if cls is SurrogateTrainer and _bf16_step(m_surrogate):
            from . import training16
            return training16.SurrogateTrainer16(recipe, m_surrogate)
        return super().__new__(cls)

    def __init__(self, recipe, m_surrogate: nn.Module):
        self.recipe, self.m = recipe, m_surrogate
        self.is_vit = hasattr(m_surrogate, "vit")
        self.n_players = _module_n_players(m_surrogate)
        self.saved = None
        self.backbone = ViTBackboneTrainer(m_surrogate.vit) if self.is_vit else BertBackboneTrainer(m_surrogate.bert)
        self.cls = Lin([m_surrogate.classifier])
        self.pool = None if self.is_vit else Lin([m_surrogate.bert_pooler.dense])
        self.step = 0

    def forward_probs(self, xs: Tensor, bits: Tensor, train: bool = True, seed: int = 0) -> Tensor:
        """fw_surrogate (probabilities [B,C]) with activations saved for ``backward_probs``."""
        cfg = self.m.config
        self.step += 1
        seeds = Seeds(seed * 104729 + self.step)
        b = xs.shape[0]
        t, h = self.n_players + 1, cfg.hidden_size
        z = self.backbone.forward(xs, bits, seeds, train)
        zc = z.view(b, t, h)[:, 0, :].contiguous()
        pooled, ph, s_pool = None, 0.0, 0
        if self.is_vit:
            logits = self.cls.forward(zc, L.AG_EPI_BIAS_F32)
        else:
            pooled = self.pool.forward(zc, L.AG_EPI_BIAS_TANH)
            s_pool = seeds.next()
            ph = cfg.hidden_dropout_prob if train else 0.0
            logits = self.cls.forward(ops.dropout(pooled, ph, s_pool), L.AG_EPI_BIAS_F32)
        probs = ops.softmax_rows(logits)
        self.saved = (b, t, h, probs, pooled, ph, s_pool)
        return probs

    def backward_probs(self, dprobs: Tensor) -> None:
        b, t, h, probs, pooled, ph, s_pool = self.saved
        self.saved = None
        dlogits = ops.softmax_rows_bwd(probs, dprobs.contiguous().float())
        if self.is_vit:
            dzc = self.cls.backward(dlogits)
        else:
            dp = ops.dropout(self.cls.backward(dlogits), ph, s_pool)
            dzc = self.pool.backward(ops.tanh_bwd(pooled, dp))
        dz = torch.zeros((b, t, h), dtype=torch.float32, device=dprobs.device)
        dz[:, 0, :].copy_(dzc)
        self.backbone.backward(dz.view(b * t, h))

    @_fp32_step
    def loss_and_grads(self, xs: Tensor, bits: Tensor, orig_probs: Tensor, train: bool = True, seed: int = 0):
        probs = self.forward_probs(xs, bits, train, seed)
        loss, dprobs = ops.kl_loss(orig_probs, probs, want_grad=True)
        self.backward_probs(dprobs)
        return loss, probs


# ------------------------------------------------------------------------------------------------ LTT (ladder side network)
class LadderTrainer:
    """LttViTModel / LttBertModel with ONE side branch (reference models/ltt_vit.py:407-440, ltt_bert.py:468-500): the
    frozen backbone is run layer by layer; after layer i the trainable ladder does
    ``side = side + gelu(map_i(hidden_i)); side = SideLayer_i(side, mask)``.  Backward touches the ladder only."""

    def __init__(self, model: nn.Module, is_vit: bool, branch: int = 0):
        self.model, self.is_vit, self.branch = model, is_vit, branch
        c = model.config
        self.kind = L.AG_MASK_VIT_MUL if is_vit else L.AG_MASK_BERT_ADD
        frozen = [model.embeddings, model.encoder.layers] + ([model.layernorm] if is_vit else [])
        if any(_any_trainable(m) for m in frozen):
            raise NotImplementedError("LTT training expects the backbone frozen (reference models/ltt_vit.py:68-74)")
        self.backbone = ViTBackboneTrainer(model) if is_vit else BertBackboneTrainer(model)
        enc = model.encoder
        n = enc.num_layers
        self.maps = [Lin([enc.s_attn_maps[f"{branch}_{i}"]]) for i in range(n)]
        self.side = [Block(enc.s_attn_layers[f"{branch}_{i}"], self.kind, c.num_attention_heads, c.layer_norm_eps,
                           c.hidden_dropout_prob, c.attention_probs_dropout_prob) for i in range(n)]
        self.ln_s = Norm(model.s_attn_layernorm[branch], c.layer_norm_eps) if is_vit else None
        self.pre: List[Tensor] = []

    def forward(self, x: Tensor, bits: Tensor, seeds: Seeds, train: bool) -> Tuple[Tensor, Tensor]:
        """-> (backbone output [B*T, H] (ViT: after the final LN), side output [B*T, h] (ViT: after its LN))."""
        enc = self.model.encoder
        b = x.shape[0]
        state = {"side": None, "t": None}
        self.pre = []

        def tap(i: int, hid: Tensor) -> None:
            if i >= enc._ltt_freeze_layer:
                return
            t = hid.shape[0] // b
            pre = self.maps[i].forward(hid)
            g = ops.gelu(pre)
            s_in = g if state["side"] is None else ops.add(state["side"], g)
            state["side"] = self.side[i].forward(s_in, bits, b, t, seeds, train)
            self.pre.append(pre)

        z = self.backbone.forward(x, bits, seeds, train, tap=tap)
        self.z_last = z                  # the frozen backbone's output (the recipes' second, non-differentiable result)
        side = state["side"]
        if self.ln_s is not None:
            side = self.ln_s.forward(side)
        return z, side

    def backward(self, dside: Tensor) -> None:
        d = self.ln_s.backward(dside) if self.ln_s is not None else dside
        for i in reversed(range(len(self.pre))):
            d = self.side[i].backward(d)                       # grad of (side_{i-1} + gelu(pre_i))
            self.maps[i].backward(ops.gelu_bwd(self.pre[i], d), need_dx=False)   # the backbone is frozen: no dX
        self.pre = []
        self.backbone.drop_saved()       # what the frozen backbone saved


class LttSurrogateTrainer:
    """LTT surrogate step (scripts/train_surrogate.py:133-147 on recipes/ltt_*): masked ladder forward, side head,
    KL against the frozen backbone's own prediction, backward into the ladder + side head."""

    def __init__(self, recipe, m_surrogate: nn.Module):
        self.recipe, self.m = recipe, m_surrogate
        self.is_vit = hasattr(m_surrogate, "vit")
        self.n_players = _module_n_players(m_surrogate)
        self.saved = None
        self.ladder = LadderTrainer(m_surrogate.vit if self.is_vit else m_surrogate.bert, self.is_vit, 0)
        self.cls = Lin([m_surrogate.s_attn_classifier])
        self.pool = None if self.is_vit else Lin([m_surrogate.bert_s_attn_pooler.dense])
        self.step = 0

    def forward_probs(self, xs: Tensor, bits: Tensor, train: bool = True, seed: int = 0) -> Tensor:
        cfg = self.m.config
        self.step += 1
        seeds = Seeds(seed * 104729 + self.step)
        b = xs.shape[0]
        t, hs = self.n_players + 1, cfg.s_attn_hidden_size
        _, side = self.ladder.forward(xs, bits, seeds, train)
        sc = side.view(b, t, hs)[:, 0, :].contiguous()
        pooled, ph, s_pool = None, 0.0, 0
        if self.is_vit:
            logits = self.cls.forward(sc, L.AG_EPI_BIAS_F32)
        else:
            pooled = self.pool.forward(sc, L.AG_EPI_BIAS_TANH)
            s_pool = seeds.next()
            ph = cfg.hidden_dropout_prob if train else 0.0
            logits = self.cls.forward(ops.dropout(pooled, ph, s_pool), L.AG_EPI_BIAS_F32)
        probs = ops.softmax_rows(logits)
        self.saved = (b, t, hs, probs, pooled, ph, s_pool)
        return probs

    def backward_probs(self, dprobs: Tensor) -> None:
        b, t, hs, probs, pooled, ph, s_pool = self.saved
        self.saved = None
        dlogits = ops.softmax_rows_bwd(probs, dprobs.contiguous().float())
        if self.is_vit:
            dsc = self.cls.backward(dlogits)
        else:
            dp = ops.dropout(self.cls.backward(dlogits), ph, s_pool)
            dsc = self.pool.backward(ops.tanh_bwd(pooled, dp))
        dside = torch.zeros((b, t, hs), dtype=torch.float32, device=dprobs.device)
        dside[:, 0, :].copy_(dsc)
        self.ladder.backward(dside.view(b * t, hs))

    @_fp32_step
    def loss_and_grads(self, xs: Tensor, bits: Tensor, orig_probs: Tensor, train: bool = True, seed: int = 0):
        probs = self.forward_probs(xs, bits, train, seed)
        loss, dprobs = ops.kl_loss(orig_probs, probs, want_grad=True)
        self.backward_probs(dprobs)
        return loss, probs


class LttExplainerTrainer:
    """LTT explainer step (scripts/train_explainer.py:182-196 on recipes/ltt_*): all-ones-mask ladder forward ->
    side explainer layers -> MLP -> normalise -> Shapley loss; backward into the ladder + side head."""

    def __init__(self, recipe, m_explainer: nn.Module):
        self.recipe, self.m = recipe, m_explainer
        cfg = m_explainer.config
        self.is_vit = hasattr(m_explainer, "vit")
        self.n_players = _module_n_players(m_explainer)
        self.saved = None
        self.kind = L.AG_MASK_VIT_MUL if self.is_vit else L.AG_MASK_BERT_ADD
        self.ladder = LadderTrainer(m_explainer.vit if self.is_vit else m_explainer.bert, self.is_vit, 0)
        layers = m_explainer.s_explainer_attn if self.is_vit else m_explainer.s_attn_attention_layers
        self.attn = [Block(ly, self.kind, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.hidden_dropout_prob,
                           cfg.attention_probs_dropout_prob) for ly in layers]
        self.mlp = MLPHead(m_explainer.s_explainer_mlp if self.is_vit else m_explainer.s_attn_explainer)
        self.step = 0

    def forward_phi(self, xs: Tensor, v_0: Optional[Tensor], v_1: Optional[Tensor], train: bool = True, seed: int = 0,
                    bits: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        cfg = self.m.config
        self.step += 1
        seeds = Seeds(seed * 7919 + self.step)
        b = xs.shape[0]
        p = self.n_players
        t, c = p + 1, cfg.num_labels
        if bits is None:
            bits = engine.ones_mask_bits(b, p, xs.device)
        _, o = self.ladder.forward(xs, bits, seeds, train)
        for blk in self.attn:
            o = blk.forward(o, bits, b, t, seeds, train)
        s_exp = seeds.next()
        ph = cfg.hidden_dropout_prob if (train and not self.is_vit) else 0.0   # BERT s_attn_exp_dropout (ltt_bert.py:205)
        o = ops.dropout(o, ph, s_exp)
        pred = self.mlp.forward(o).view(b, t, c)
        phi = ops.shapley_normalize(pred, v_1, v_0, normalize=bool(cfg.explainer_normalize))
        self.saved = (b, t, c, ph, s_exp)
        return phi, None

    def backward_phi(self, dphi: Tensor, dbase: Optional[Tensor] = None) -> None:
        cfg = self.m.config
        b, t, c, ph, s_exp = self.saved
        self.saved = None
        dpred = ops.shapley_normalize_bwd(dphi, t, normalize=bool(cfg.explainer_normalize)).view(b * t, c)
        d = ops.dropout(self.mlp.backward(dpred), ph, s_exp)
        for blk in reversed(self.attn):
            d = blk.backward(d)
        self.ladder.backward(d)

    @_fp32_step
    def loss_and_grads(self, xs: Tensor, bits_loss: Tensor, v_0: Tensor, v_s: Tensor, v_1: Tensor, n_mask_samples: int,
                       labels: Optional[Tensor] = None, train: bool = True, seed: int = 0):
        phi, _ = self.forward_phi(xs, v_0, v_1, train, seed)
        loss, dphi = ops.shapley_loss(bits_loss, v_0, v_s, phi, xs.shape[0], n_mask_samples, want_grad=True)
        self.backward_phi(dphi)
        return loss, phi


def _is_ltt(recipe, module: nn.Module) -> bool:
    return recipe.id.startswith("ltt_") if recipe is not None else type(module).__name__.startswith("Ltt")


def make_explainer_trainer(recipe, m_explainer: nn.Module):
    return LttExplainerTrainer(recipe, m_explainer) if _is_ltt(recipe, m_explainer) else ExplainerTrainer(recipe, m_explainer)


def make_surrogate_trainer(recipe, m_surrogate: nn.Module):
    return LttSurrogateTrainer(recipe, m_surrogate) if _is_ltt(recipe, m_surrogate) else SurrogateTrainer(recipe, m_surrogate)
