"""The bf16 training step (``training.MIXED_BF16``) of the vanilla / duo / froyo explainers and surrogates, round 4.

What torch.autograd does for the reference in scripts/train_explainer.py:183-198, scripts/train_duo_explainer.py:180-198 and
scripts/train_surrogate.py:143-147, restated for under-filled launches (B*T ~ 1-1.6 k token rows per GPU):

* every Linear forward / dX / dW is ONE ``ag_gemm_ex`` launch that reads its operands in place (NT / NN / TN orders): no transposed
  or re-cast copy of an activation, a gradient or a weight exists; products with few output tiles are split over the contraction
  into fp32 slabs that the next row kernel adds in slab order;
* activations are saved ONCE, as the bf16 operand the forward GEMM read (which is also what the dW GEMM reads); the residual stream,
  LayerNorm statistics, soft-max and every gradient that is accumulated stay fp32;
* everything between two Linear layers is one row kernel: bias + dropout + residual + LayerNorm forward (``ag_rows_finish``),
  LayerNorm backward + residual-branch gradient + the bf16 dY operand and the bias gradient of the Linear below
  (``ag_rows_ln_bwd``); GELU is an epilogue of fc1 (forward) and of fc2's dX (backward);
* the dW products are off the dependency chain of the backward: they run on a second HIP stream beside the dX chain and are
  joined once, before the optimiser;
* the bf16 forms of all weights are refreshed by one launch after an optimiser step (``ag_cast_f32_many``).

Same trainer interface as ``training.py`` (forward_phi / backward_phi / loss_and_grads ...): ``training.ExplainerTrainer`` and
``training.SurrogateTrainer`` delegate here when MIXED_BF16 is on and the model has the shapes the kernels cover (head dim 64,
hidden size a multiple of 8 up to 1024, T <= 256); the exact-fp32 step and the LTT ladder stay in ``training.py``.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _lib as L
from . import engine, ops
from . import training as T

BF16, F32 = L.AG_BF16, L.AG_F32
VIT, BERT = L.AG_MASK_VIT_MUL, L.AG_MASK_BERT_ADD
SIDE_STREAM = os.environ.get("AG_TRAIN_SIDE", "1") != "0"
N_SIDE = max(1, int(os.environ.get("AG_TRAIN_SIDE", "1") or 1))     # number of side streams (AG_TRAIN_SIDE=0: none)
SIDE_MIN_ROWS = int(os.environ.get("AG_TRAIN_SIDE_MIN_ROWS", "1024"))   # token rows of a step below which the dW products stay on the main stream
# hipGraph capture of the explainer step (forward + loss + backward, both streams): "1" on, "0" off (the default: the epoch bodies
# keep the GPU busy with the K-mask target forward while the host issues the step, so eager launches cost nothing there; it pays at
# 2-4 images per step, where the ~420 launches are host-bound).  Never used while a gradient sink is installed (N > 1 ranks: the
# bucket reducer must be told, kernel by kernel, which gradients are final).
GRAPH_STEP = os.environ.get("AG_TRAIN_GRAPH", "0") == "1"
# dW products per grouped launch (round 6; _Side.defer_dw): 8 = the Linears of two layers; 0 / 1: every Linear's dW as its own
# ag_gemm_ex launch (+ slab reduction), the round-4 form (A/B)
DW_GROUP = int(os.environ.get("AG_TRAIN_DW_GROUP", "8"))


def supported(module: nn.Module) -> bool:
    """shapes the bf16 step covers (else the fp32-activation step of training.py runs)."""
    cfg = module.config
    h, heads = cfg.hidden_size, cfg.num_attention_heads
    t = T._module_n_players(module) + 1
    return h == heads * 64 and h % 8 == 0 and h <= 1024 and cfg.intermediate_size % 8 == 0 and t <= 256


# ------------------------------------------------------------------------------------------------ side stream
class _Side:
    """The side streams of the backward.  ``run(fn, keep)`` forks from the current stream (everything issued so far is visible to
    fn) onto the next of ``N_SIDE`` streams (round robin: the dW products of one layer are independent of each other as well),
    ``join()`` makes the current stream wait for all forked work.  ``keep``: tensors allocated on the main stream that fn reads —
    held until the join so that the caching allocator cannot hand their memory to a later main-stream kernel."""

    _per_device = {}

    def __init__(self, device):
        self.streams = [torch.cuda.Stream(device) for _ in range(N_SIDE)] if SIDE_STREAM else []
        self.turn = 0
        self.keep: List[Tensor] = []
        self.pending: List[Tuple] = []     # deferred dW products (defer_dw)
        self.finals: List[Tensor] = []     # parameters whose gradient was produced on the MAIN stream since the last fork
        self.dirty = False
        # this backward runs its dW products in line (set per step by the trainers): under hipGraph capture — a one-branch graph replays at
        # the eager rate whatever GPU_MAX_HW_QUEUES is, a forked one at half of it with 8 queues (vanilla ViT-base 524 vs 353 images/s) —
        # and below SIDE_MIN_ROWS token rows, where fork and join cost more than the products they move (duo BERT at 2 images: 259 -> 337)
        self.inline = False

    @classmethod
    def of(cls, device) -> "_Side":
        key = str(device)
        if key not in cls._per_device:
            cls._per_device[key] = cls(device)
        return cls._per_device[key]

    def run(self, fn, *keep: Tensor) -> None:
        if not self.streams or self.inline:
            fn()
            self._report()
            return
        # (N > 1 ranks: the gradient sink buckets what it is told in report order and reduces a bucket on the reporting stream —
        # all gradients then come from ONE side stream, so a bucket never reads a gradient another side stream is still writing)
        st = self.streams[0] if T.GRAD_SINK is not None else self.streams[self.turn]
        self.turn = (self.turn + 1) % len(self.streams)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
            if st is self.streams[0]:
                self._report()      # (gradients finished on the main stream before this fork are ordered before it too)
        self.keep.extend(keep)
        self.dirty = True

    def defer_dw(self, lw, dyb: Tensor, xb: Tensor, db: Optional[Tensor]) -> None:
        """(round 6) the dW product of one Linear joins the pending group; DW_GROUP of them (two layers' Linears) leave as ONE
        ag_gemm_ex_group launch + ONE grouped column-sum launch on the side stream.  Alone, a dW product of 36-144 tiles needed 3-6
        contraction ranges to cover the chip and a slab reduction behind it (two or three launches of ~15 us per Linear, ten per layer);
        eight of them are ~860 tiles: no contraction ranges, no reduction, two launches per two layers — and 64 MiB of gradients, one
        exchange bucket (distributed.GradBucketReducer), become final together."""
        self.pending.append((lw, dyb, xb, db))
        if len(self.pending) >= DW_GROUP:
            self.flush_dw()

    def flush_dw(self) -> None:
        if not self.pending:
            return
        group, self.pending = self.pending, []

        def work():
            gs = ops.gemm_ex_group([(dyb, xb) for _, dyb, xb, _ in group], ops.TN, F32)
            sums = iter(ops.colsum_bf16_group([dyb for _, dyb, _, db in group if db is None]))
            for (lw, dyb, xb, db), g in zip(group, gs):
                bias_g = db if db is not None else next(sums)
                off = 0
                for mod in lw.mods:
                    rows = mod.weight.shape[0]
                    if mod.weight.requires_grad:
                        T._acc_grad(mod.weight, g[off:off + rows], fresh=True)
                    if mod.bias.requires_grad:
                        T._acc_grad(mod.bias, bias_g[off:off + rows], fresh=True)
                    off += rows
        keep = [t_ for _, dyb, xb, db in group for t_ in (dyb, xb, db) if t_ is not None]
        self.run(work, *keep)

    def begin(self, inline: bool) -> None:
        """start of a backward.  A backward that raised half way (an allocation failure, a bad shape) leaves deferred products, kept
        tensors and unreported gradients of THAT step behind: they must not leave with this step's first group."""
        self.pending.clear()
        self.finals.clear()
        if self.dirty:                     # (forked work of the aborted step may still read what ``keep`` holds)
            cur = torch.cuda.current_stream()
            for st in self.streams:
                cur.wait_stream(st)
            self.dirty = False
        self.keep.clear()
        self.turn = 0
        self.inline = inline

    def final_on_main(self, *params: Tensor) -> None:
        """a gradient that a main-stream kernel wrote: reported to training.GRAD_SINK at a later fork (or at the join), on a
        stream that is ordered behind it."""
        self.finals.extend(p for p in params if p.requires_grad)

    def _report(self) -> None:
        if self.finals:
            fin, self.finals = self.finals, []
            T._final(*fin)

    def join(self) -> None:
        self.flush_dw()
        if self.dirty:
            cur = torch.cuda.current_stream()
            for st in self.streams:
                cur.wait_stream(st)
        self.keep.clear()
        self.dirty = False
        self.turn = 0
        self._report()


# ------------------------------------------------------------------------------------------------ weights
class LinW:
    """bf16 [rows_padded, K] weight and fp32 [rows_padded] bias of one (fused) Linear, refreshed by the bank."""
    __slots__ = ("mods", "w", "b", "n", "k", "key")

    def __init__(self, mods: Sequence[nn.Module], pad_rows_to: int, device):
        self.mods = list(mods)
        self.n = sum(m.weight.shape[0] for m in self.mods)
        self.k = self.mods[0].weight[0].numel()
        rows = (self.n + pad_rows_to - 1) // pad_rows_to * pad_rows_to
        self.w = torch.zeros((rows, self.k), dtype=torch.bfloat16, device=device)
        self.b = torch.zeros(rows, dtype=torch.float32, device=device)
        self.key = None

    @property
    def trainable(self) -> bool:
        return any(p.requires_grad for m in self.mods for p in (m.weight, m.bias))


class WeightBank:
    """bf16 operand forms of every Linear a trainer touches.  ``refresh()``: ONE ag_cast_f32_many launch for the Linears whose
    parameters changed since the last call (engine.param_key: optimiser steps, loads) — all of them after an optimiser step, none
    for frozen modules after the first step."""

    def __init__(self, device):
        self.device = device
        self.lins: List[LinW] = []

    def linear(self, mods: Sequence[nn.Module], pad_rows_to: int = 1) -> LinW:
        lw = LinW(mods, pad_rows_to, self.device)
        self.lins.append(lw)
        return lw

    def keys(self, trainable: bool):
        """current parameter keys of the trainable (or the frozen) Linears: what a captured step compares to notice a reload."""
        return tuple(tuple(engine.param_key(p) for m in lw.mods for p in (m.weight, m.bias)) for lw in self.lins if lw.trainable == trainable)

    def invalidate(self, trainable: bool) -> None:
        for lw in self.lins:
            if lw.trainable == trainable:
                lw.key = None

    def refresh(self) -> None:
        pairs = []
        for lw in self.lins:
            key = tuple(engine.param_key(p) for m in lw.mods for p in (m.weight, m.bias))
            if key == lw.key:
                continue
            off = 0
            for m in lw.mods:
                rows = m.weight.shape[0]
                pairs.append((m.weight.detach().reshape(rows, -1), lw.w[off:off + rows]))
                pairs.append((m.bias.detach(), lw.b[off:off + rows]))
                off += rows
            lw.key = key
        if pairs:
            ops.cast_many(pairs)


def _ln(mod: nn.Module, eps: float) -> Optional[Tuple[Tensor, Tensor, float]]:
    """(gamma, beta, eps) of a LayerNorm module for ag_rows_finish; None for nn.Identity."""
    if isinstance(mod, nn.Identity):
        return None
    return (mod.weight.detach(), mod.bias.detach(), eps)


def _ln_grads(mod: nn.Module, dev) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    if isinstance(mod, nn.Identity) or not mod.weight.requires_grad:
        return None, None
    h = mod.weight.shape[0]
    return torch.empty(h, dtype=torch.float32, device=dev), torch.empty(h, dtype=torch.float32, device=dev)


def _give_ln_grads(side: _Side, mod: nn.Module, dg: Optional[Tensor], db: Optional[Tensor]) -> None:
    if dg is None:
        return
    _give(mod.weight, dg)
    _give(mod.bias, db)
    side.final_on_main(mod.weight, mod.bias)


def _give(p: Tensor, src: Tensor) -> None:
    """p.grad (+)= src WITHOUT reporting it final (the caller reports through the side stream protocol)."""
    src = src.reshape(p.shape)
    T._touch(p)
    if p.grad is None:
        p.grad = src if src.is_contiguous() else src.contiguous()
    else:
        T._acc(p.grad, src)


class Lin16:
    """One (fused) Linear on ag_gemm_ex: forward NT, dX NN, dW TN — operands read in place."""

    def __init__(self, bank: WeightBank, mods: Sequence[nn.Module], pad_rows_to: int = 1):
        self.lw = bank.linear(mods, pad_rows_to)

    @property
    def w(self) -> Tensor:
        return self.lw.w

    @property
    def b(self) -> Tensor:
        return self.lw.b

    @property
    def trainable(self) -> bool:
        return self.lw.trainable

    def fwd(self, xb: Tensor, out_dtype: int = BF16) -> Tensor:
        return ops.gemm_ex(xb, self.w, ops.NT, L.AG_EX_STORE, bias=self.b, out_dtype=out_dtype)

    def fwd_slabs(self, xb: Tensor) -> Tensor:
        """x W^T as split-K slabs [S, M, N] (the bias joins in the row kernel that adds them)."""
        return ops.gemm_ex(xb, self.w, ops.NT, L.AG_EX_SLABS)

    def fwd_gelu(self, xb: Tensor) -> Tuple[Tensor, Tensor]:
        return ops.gemm_ex(xb, self.w, ops.NT, L.AG_EX_GELU_DUAL, bias=self.b)

    def dx_slabs(self, dyb: Tensor) -> Tensor:
        return ops.gemm_ex(dyb, self.w, ops.NN, L.AG_EX_SLABS)

    def dx_gelu(self, dyb: Tensor, pre: Tensor) -> Tensor:
        """(dY W) * gelu'(pre): the gradient wrt the pre-activation of the Linear that fed this one through a GELU."""
        return ops.gemm_ex(dyb, self.w, ops.NN, L.AG_EX_GELU_BWD, aux=pre)

    def dw(self, side: _Side, dyb: Tensor, xb: Tensor, db: Optional[Tensor]) -> None:
        """dW = dY^T X (+ the bias gradient: ``db`` given, or the column sums of dyb) into param.grad, on the side stream."""
        if not self.trainable:
            return
        lw = self.lw
        if DW_GROUP > 1:
            side.defer_dw(lw, dyb, xb, db)
            return

        def work():
            n, k = dyb.shape[1], xb.shape[1]
            splits = ops.gemm_ex_splits(n, k, dyb.shape[0])
            if splits == 1:
                g = ops.gemm_ex(dyb, xb, ops.TN, L.AG_EX_STORE, out_dtype=F32)
            else:
                g = ops.slab_reduce(ops.gemm_ex(dyb, xb, ops.TN, L.AG_EX_SLABS, splits=splits))
            bias_g = db if db is not None else ops.colsum_bf16(dyb)
            off = 0
            for mod in lw.mods:
                rows = mod.weight.shape[0]
                if mod.weight.requires_grad:
                    T._acc_grad(mod.weight, g[off:off + rows], fresh=True)
                if mod.bias.requires_grad:
                    T._acc_grad(mod.bias, bias_g[off:off + rows], fresh=True)
                off += rows
        side.run(work, dyb, xb, *( [db] if db is not None else []))


class _Below:
    """the dropout the consumer BELOW a block applies to this block's input gradient before its own fc2 backward."""
    __slots__ = ("p", "seed", "want")

    def __init__(self, p: float = 0.0, seed: int = 0, want: bool = False):
        self.p, self.seed, self.want = p, seed, want


NO_BELOW = _Below()


# ------------------------------------------------------------------------------------------------ blocks
class ViTBlock16:
    """Pre-LN layer (reference VanillaViTLayer, models/vanilla_vit.py:364-377).  Carries the residual stream h (fp32) and the bf16
    operand u = LN_before(h) the block starts from; its last row kernel already applies the NEXT norm (the following block's
    layernorm_before, the backbone's final LayerNorm, the explainer head's LayerNorm, or none)."""

    def __init__(self, bank: WeightBank, layer: nn.Module, heads: int, eps: float, p_hidden: float, p_attn: float):
        att = layer.attention
        self.heads, self.eps, self.p_hidden, self.p_attn = heads, eps, p_hidden, p_attn
        self.qkv = Lin16(bank, [att.self.query, att.self.key, att.self.value])
        self.o, self.fc1, self.fc2 = Lin16(bank, [att.output.dense]), Lin16(bank, [layer.intermediate.dense]), Lin16(bank, [layer.output.dense])
        self.n1, self.n2 = layer.layernorm_before, layer.layernorm_after
        self.trainable = T._any_trainable(layer)
        self.saved = None
        self.ph = 0.0
        self.s_f = 0

    def forward(self, h: Tensor, ub: Tensor, bits: Tensor, rows: int, t: int, seeds: T.Seeds, train: bool, next_ln, save: bool,
                want_f32: bool = False) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        """-> (h_out fp32, bf16(next_norm(h_out)), fp32 next_norm(h_out) if want_f32)."""
        hd = h.shape[-1]
        ph, pa = (self.p_hidden, self.p_attn) if train else (0.0, 0.0)
        s_att, s_o, s_f = seeds.next(), seeds.next(), seeds.next()
        qkv = self.qkv.fwd(ub)
        ctx = ops.masked_attention_train_bf16(qkv, bits, rows, t, hd, self.heads, VIT, pa, s_att)
        hx, _, u2b = ops.rows_finish(self.o.fwd_slabs(ctx), bias=self.o.b, p_drop=ph, seed=s_o, resid=h, ln=_ln(self.n2, self.eps), want_t=True)
        f1b, gb = self.fc1.fwd_gelu(u2b)
        h2, zf, zb = ops.rows_finish(self.fc2.fwd_slabs(gb), bias=self.fc2.b, p_drop=ph, seed=s_f, resid=hx, ln=next_ln, want_t=True,
                                     want_f32=want_f32)
        self.ph, self.s_f = ph, s_f
        self.saved = (h, ub, qkv, ctx, hx, u2b, f1b, gb, bits, rows, t, ph, pa, s_att, s_o) if save else None
        return h2, zb, zf

    def backward(self, side: _Side, dout: Tensor, dyb: Tensor, db2: Optional[Tensor], below: _Below, need_dx: bool = True):
        """dout = d loss / d h_out (fp32); dyb = bf16(dropout'(dout; ph, s_f)) with db2 its column sums (made by the producer of
        dout in the same pass) -> (d loss / d h_in, its bf16 dropout'(below) form, column sums of that) or None."""
        h, ub, qkv, ctx, hx, u2b, f1b, gb, bits, rows, t, ph, pa, s_att, s_o = self.saved
        self.saved = None
        hd, dev = h.shape[-1], h.device
        self.fc2.dw(side, dyb, gb, db2)
        df1b = self.fc2.dx_gelu(dyb, f1b)
        self.fc1.dw(side, df1b, u2b, None)
        dg2, db2n = _ln_grads(self.n2, dev)
        dbo = torch.empty(hd, dtype=torch.float32, device=dev) if self.o.trainable else None
        dhx, dyo = ops.rows_ln_bwd(self.fc1.dx_slabs(df1b), x=hx, gamma=self.n2.weight.detach(), eps=self.eps, add=dout, want_bf16=True,
                                   p_drop=ph, seed=s_o, dgamma=dg2, dbeta=db2n, dbias=dbo)
        _give_ln_grads(side, self.n2, dg2, db2n)
        self.o.dw(side, dyo, ctx, dbo)
        dqkvb = ops.masked_attention_bwd_bf16(qkv, bits, self.o.dx_slabs(dyo), rows, t, hd, self.heads, VIT, pa, s_att)
        self.qkv.dw(side, dqkvb, ub, None)
        ident = isinstance(self.n1, nn.Identity)
        if not need_dx and (ident or not self.n1.weight.requires_grad):
            return None
        dg1, db1n = _ln_grads(self.n1, dev)
        dbb = torch.empty(hd, dtype=torch.float32, device=dev) if below.want else None
        dh, dyb_below = ops.rows_ln_bwd(self.qkv.dx_slabs(dqkvb), x=None if ident else h, gamma=None if ident else self.n1.weight.detach(),
                                        eps=self.eps, add=dhx, want_bf16=below.want, p_drop=below.p, seed=below.seed, dgamma=dg1,
                                        dbeta=db1n, dbias=dbb)
        _give_ln_grads(side, self.n1, dg1, db1n)
        return dh, dyb_below, dbb


class BertBlock16:
    """Post-LN layer (reference VanillaBertLayer, models/vanilla_bert.py:410-427, :556-560, :600-604).  Carries the post-LayerNorm
    stream x (fp32) and its bf16 form; its input gradient leaves as un-added pieces (split-K slabs of the QKV dX + the residual
    branch) that the row kernel of the layer below adds."""

    def __init__(self, bank: WeightBank, layer: nn.Module, heads: int, eps: float, p_hidden: float, p_attn: float):
        att = layer.attention
        self.heads, self.eps, self.p_hidden, self.p_attn = heads, eps, p_hidden, p_attn
        self.qkv = Lin16(bank, [att.self.query, att.self.key, att.self.value])
        self.o, self.fc1, self.fc2 = Lin16(bank, [att.output.dense]), Lin16(bank, [layer.intermediate.dense]), Lin16(bank, [layer.output.dense])
        self.n1, self.n2 = att.output.LayerNorm, layer.output.LayerNorm
        self.trainable = T._any_trainable(layer)
        self.saved = None

    def forward(self, x0: Tensor, x0b: Tensor, bits: Tensor, rows: int, t: int, seeds: T.Seeds, train: bool, save: bool) -> Tuple[Tensor, Tensor]:
        hd = x0.shape[-1]
        ph, pa = (self.p_hidden, self.p_attn) if train else (0.0, 0.0)
        s_att, s_o, s_f = seeds.next(), seeds.next(), seeds.next()
        qkv = self.qkv.fwd(x0b)
        ctx = ops.masked_attention_train_bf16(qkv, bits, rows, t, hd, self.heads, BERT, pa, s_att)
        ident = isinstance(self.n1, nn.Identity)
        t1, a, ab = ops.rows_finish(self.o.fwd_slabs(ctx), bias=self.o.b, p_drop=ph, seed=s_o, resid=x0, ln=_ln(self.n1, self.eps),
                                    want_t=not ident, want_f32=True)
        f1b, gb = self.fc1.fwd_gelu(ab)
        t2, x1, x1b = ops.rows_finish(self.fc2.fwd_slabs(gb), bias=self.fc2.b, p_drop=ph, seed=s_f, resid=a, ln=_ln(self.n2, self.eps),
                                      want_t=True, want_f32=True)
        self.saved = (x0b, qkv, ctx, t1, ab, f1b, gb, t2, bits, rows, t, ph, pa, s_att, s_o, s_f) if save else None
        return x1, x1b

    def backward(self, side: _Side, dy: Tensor, dy_add: Optional[Tensor], need_dx: bool = True):
        """d loss / d x_out = sum(dy slabs) + dy_add  ->  (slabs, add) of d loss / d x_in, or None."""
        x0b, qkv, ctx, t1, ab, f1b, gb, t2, bits, rows, t, ph, pa, s_att, s_o, s_f = self.saved
        self.saved = None
        hd, dev = qkv.shape[-1] // 3, qkv.device
        dg2, db2n = _ln_grads(self.n2, dev)
        db2 = torch.empty(hd, dtype=torch.float32, device=dev) if self.fc2.trainable else None
        dt2, dyb = ops.rows_ln_bwd(dy, x=t2, gamma=self.n2.weight.detach(), eps=self.eps, dy_add=dy_add, want_bf16=True, p_drop=ph, seed=s_f,
                                   dgamma=dg2, dbeta=db2n, dbias=db2)
        _give_ln_grads(side, self.n2, dg2, db2n)
        self.fc2.dw(side, dyb, gb, db2)
        df1b = self.fc2.dx_gelu(dyb, f1b)
        self.fc1.dw(side, df1b, ab, None)
        ident = isinstance(self.n1, nn.Identity)
        dg1, db1n = _ln_grads(self.n1, dev)
        dbo = torch.empty(hd, dtype=torch.float32, device=dev) if self.o.trainable else None
        dt1, dyo = ops.rows_ln_bwd(self.fc1.dx_slabs(df1b), x=None if ident else t1, gamma=None if ident else self.n1.weight.detach(),
                                   eps=self.eps, dy_add=dt2, want_bf16=True, p_drop=ph, seed=s_o, dgamma=dg1, dbeta=db1n, dbias=dbo)
        _give_ln_grads(side, self.n1, dg1, db1n)
        self.o.dw(side, dyo, ctx, dbo)
        dqkvb = ops.masked_attention_bwd_bf16(qkv, bits, self.o.dx_slabs(dyo), rows, t, hd, self.heads, BERT, pa, s_att)
        self.qkv.dw(side, dqkvb, x0b, None)
        if not need_dx:
            return None
        return self.qkv.dx_slabs(dqkvb), dt1


# ------------------------------------------------------------------------------------------------ backbones
class ViTBackbone16:
    """VanillaViTModel (embeddings + encoder + final LayerNorm), reference models/vanilla_vit.py:207-214."""

    def __init__(self, bank: WeightBank, vit: nn.Module):
        self.vit, c = vit, vit.config
        self.frozen = not T._any_trainable(vit)
        self.blocks = [ViTBlock16(bank, ly, c.num_attention_heads, c.layer_norm_eps, c.hidden_dropout_prob, c.attention_probs_dropout_prob)
                       for ly in vit.encoder.layers]
        self.proj = Lin16(bank, [vit.embeddings.patch_embeddings.projection])
        self.saved = None

    def forward(self, x: Tensor, bits: Tensor, seeds: T.Seeds, train: bool) -> Tuple[Tensor, Tensor]:
        """-> (z fp32, z bf16) with z = LN_final(hidden) [B*T, H]."""
        c, e = self.vit.config, self.vit.embeddings
        b, p, h = x.shape[0], self.vit.n_players, c.hidden_size
        t = p + 1
        save = not self.frozen
        x = x.contiguous().float()
        dev = x.device
        cols = torch.empty((b * p, c.img_channels * c.img_patch_size ** 2), dtype=torch.bfloat16, device=dev)
        h0 = torch.empty((b * t, h), dtype=torch.float32, device=dev)
        with L.on(dev):
            L.check(L.lib().ag_vit_im2col(L.ptr(x), b, c.img_channels, c.img_px_size, c.img_patch_size, L.ptr(cols), BF16, L.stream()))
            pe = self.proj.fwd(cols, F32)
            L.check(L.lib().ag_vit_assemble(L.ptr(pe), L.ptr(e.cls_token.detach().float().contiguous()),
                                            L.ptr(e.position_embeddings.detach().float().contiguous()), b, p, h, L.ptr(h0), L.stream()))
        s_emb = seeds.next()
        ph = c.hidden_dropout_prob if train else 0.0
        eps = c.layer_norm_eps
        hid, _, ub = ops.rows_finish(h0, p_drop=ph, seed=s_emb, ln=_ln(self.blocks[0].n1, eps), want_t=ph > 0.0)
        hid = h0 if hid is None else hid
        n = len(self.blocks)
        zf = None
        for i, blk in enumerate(self.blocks):
            nxt = _ln(self.blocks[i + 1].n1, eps) if i + 1 < n else _ln(self.vit.layernorm, eps)
            hid, ub, zf = blk.forward(hid, ub, bits, b, t, seeds, train, nxt, save, want_f32=(i + 1 == n))
        self.saved = (b, p, h, ph, s_emb, cols, hid) if save else None
        return zf, ub

    def backward(self, side: _Side, dz: Tensor, dz_add: Optional[Tensor] = None) -> None:
        b, p, h, ph, s_emb, cols, h_last = self.saved
        self.saved = None
        t = p + 1
        dev = dz.device
        top = self.blocks[-1]
        lnf = self.vit.layernorm
        dgf, dbf = _ln_grads(lnf, dev)
        db2 = torch.empty(h, dtype=torch.float32, device=dev)
        d, dyb = ops.rows_ln_bwd(dz, x=h_last, gamma=lnf.weight.detach(), eps=self.vit.config.layer_norm_eps, dy_add=dz_add, want_bf16=True,
                                 p_drop=top.ph, seed=top.s_f, dgamma=dgf, dbeta=dbf, dbias=db2)
        _give_ln_grads(side, lnf, dgf, dbf)
        for i in reversed(range(len(self.blocks))):
            below = _Below(self.blocks[i - 1].ph, self.blocks[i - 1].s_f, True) if i > 0 else NO_BELOW
            d, dyb, db2 = self.blocks[i].backward(side, d, dyb, db2, below)
        d = ops.dropout(d, ph, s_emb).view(b, t, h)
        e = self.vit.embeddings
        if e.position_embeddings.requires_grad:
            _give(e.position_embeddings, ops.colsum(d.reshape(b, t * h)))
            side.final_on_main(e.position_embeddings)
        if e.cls_token.requires_grad:
            _give(e.cls_token, ops.colsum(d[:, 0, :].contiguous()))
            side.final_on_main(e.cls_token)
        if self.proj.trainable:
            dpb = ops.cast(d[:, 1:, :].contiguous().view(b * p, h), BF16)
            self.proj.dw(side, dpb, cols, None)


class BertBackbone16:
    """VanillaBertModel (embeddings + LayerNorm + encoder), reference models/vanilla_bert.py:248-268."""

    def __init__(self, bank: WeightBank, bert: nn.Module):
        self.bert, c = bert, bert.config
        self.frozen = not T._any_trainable(bert)
        self.blocks = [BertBlock16(bank, ly, c.num_attention_heads, c.layer_norm_eps, c.hidden_dropout_prob, c.attention_probs_dropout_prob)
                       for ly in bert.encoder.layers]
        self.ln_e = T.Norm(bert.embeddings.LayerNorm, c.layer_norm_eps)
        self.saved = None

    def forward(self, ids: Tensor, bits: Tensor, seeds: T.Seeds, train: bool) -> Tuple[Tensor, Tensor]:
        c, e = self.bert.config, self.bert.embeddings
        ids = ids.contiguous().to(torch.int64)
        b, t = ids.shape
        h = c.hidden_size
        save = not self.frozen
        # gather (index plumbing) + adds through the add kernel, then LayerNorm — as training.BertBackboneTrainer
        emb = e.word_embeddings.weight.detach().float()[ids].view(b * t, h).contiguous()
        pos_type = ops.add(e.position_embeddings.weight.detach().float()[:t].contiguous(),
                           e.token_type_embeddings.weight.detach().float()[0:1].expand(t, h).contiguous())
        emb = ops.add(emb, pos_type.repeat(b, 1).contiguous())
        s_emb = seeds.next()
        ph = c.hidden_dropout_prob if train else 0.0
        y = self.ln_e.forward(emb)
        hid, _, xb = ops.rows_finish(y, p_drop=ph, seed=s_emb, want_t=ph > 0.0)
        x = y if hid is None else hid
        for blk in self.blocks:
            x, xb = blk.forward(x, xb, bits, b, t, seeds, train, save)
        if not save:
            self.ln_e.x = None
        self.saved = (ids, b, t, h, ph, s_emb) if save else None
        return x, xb

    def backward(self, side: _Side, dy: Tensor, dy_add: Optional[Tensor] = None) -> None:
        ids, b, t, h, ph, s_emb = self.saved
        self.saved = None
        for blk in reversed(self.blocks):
            dy, dy_add = blk.backward(side, dy, dy_add)
        # embeddings LayerNorm backward on the row kernel (block-order partials: bit-reproducible, unlike the LDS atomics of
        # ag_layernorm_bwd); the embedding dropout sits between the encoder and that LayerNorm
        lne, emb = self.ln_e.mod, self.ln_e.x
        self.ln_e.x = None
        dge, dbe = _ln_grads(lne, dy.device)
        if ph > 0.0:
            d, _ = ops.rows_ln_bwd(dy, dy_add=dy_add)
            dy, dy_add = ops.dropout(d, ph, s_emb), None
        d, _ = ops.rows_ln_bwd(dy, x=emb, gamma=lne.weight.detach(), eps=self.ln_e.eps, dy_add=dy_add, dgamma=dge, dbeta=dbe)
        _give_ln_grads(side, lne, dge, dbe)
        e = self.bert.embeddings
        if e.position_embeddings.weight.requires_grad:
            g = T._grad(e.position_embeddings.weight)
            T._acc(g[:t], ops.colsum(d.view(b, t * h)).view(t, h))
            side.final_on_main(e.position_embeddings.weight)
        if e.token_type_embeddings.weight.requires_grad:
            g = T._grad(e.token_type_embeddings.weight)
            T._acc(g[0], ops.colsum(d))
            side.final_on_main(e.token_type_embeddings.weight)
        if e.word_embeddings.weight.requires_grad:
            # scatter-add of B*T rows into the vocabulary table: index plumbing; [PAD] rows get no gradient
            # (nn.Embedding(padding_idx), reference models/vanilla_bert.py:288-290)
            pad = e.word_embeddings.padding_idx
            flat = ids.view(-1)
            if pad is not None:
                d = d.masked_fill((flat == pad).unsqueeze(1), 0.0)
            T._grad(e.word_embeddings.weight).index_add_(0, flat, d)
            side.final_on_main(e.word_embeddings.weight)


# ------------------------------------------------------------------------------------------------ explainer head
CPAD = 16   # the C-wide last Linear (C = 10 / 2 classes) is padded to 16 columns for ag_gemm_ex


class MLPHead16:
    """explainer_mlp: ViT [LN, Linear, GELU, Linear, GELU, Linear] (models/vanilla_vit.py:92-100) / BERT without the LN
    (models/vanilla_bert.py:114-121).  The LayerNorm is applied by the row kernel that produced the head's input."""

    def __init__(self, bank: WeightBank, seq: nn.Sequential):
        mods = list(seq)
        self.ln = mods[0] if isinstance(mods[0], nn.LayerNorm) else None
        lins = [m for m in mods if isinstance(m, nn.Linear)]
        self.l1, self.l2 = Lin16(bank, [lins[0]]), Lin16(bank, [lins[1]])
        self.l3 = Lin16(bank, [lins[2]], pad_rows_to=CPAD)
        self.c = lins[2].weight.shape[0]
        self.saved = None

    def next_ln(self):
        return None if self.ln is None else (self.ln.weight.detach(), self.ln.bias.detach(), self.ln.eps)

    def forward(self, xb: Tensor) -> Tensor:
        """xb = bf16(LN(o)) -> pred fp32 [M, C]."""
        a_b, ga_b = self.l1.fwd_gelu(xb)
        b_b, gb_b = self.l2.fwd_gelu(ga_b)
        pred = ops.pad_cols(self.l3.fwd(gb_b, F32), self.c, F32) if self.l3.w.shape[0] != self.c else self.l3.fwd(gb_b, F32)
        self.saved = (xb, a_b, ga_b, b_b, gb_b)
        return pred

    def backward(self, side: _Side, dpred: Tensor) -> Tensor:
        """-> split-K slabs of d loss / d (head input after its LayerNorm)."""
        xb, a_b, ga_b, b_b, gb_b = self.saved
        self.saved = None
        dpp = ops.pad_cols(dpred, self.l3.w.shape[0], BF16)
        db3 = ops.colsum(dpred) if self.l3.trainable else None
        if self.l3.trainable:
            lw, c = self.l3.lw, self.c

            def work():
                g = ops.gemm_ex(dpp, gb_b, ops.TN, L.AG_EX_STORE, out_dtype=F32)    # [CPAD, Hh]; rows >= C are zero products
                m = lw.mods[0]
                if m.weight.requires_grad:
                    T._acc_grad(m.weight, g[:c], fresh=True)
                if m.bias.requires_grad:
                    T._acc_grad(m.bias, db3, fresh=True)
            side.run(work, dpp, gb_b, db3)
        d_b = self.l3.dx_gelu(dpp, b_b)
        self.l2.dw(side, d_b, ga_b, None)
        d_a = self.l2.dx_gelu(d_b, a_b)
        self.l1.dw(side, d_a, xb, None)
        return self.l1.dx_slabs(d_a)


def _cls_rows(z: Tensor, b: int, t: int, h: int) -> Tensor:
    return z.view(b, t, h)[:, 0, :].contiguous()


# ------------------------------------------------------------------------------------------------ captured step
class _StepGraph:
    """One hipGraph of ``trainer._loss_and_grads_eager`` (weight refresh + forward + loss + backward on the main and the side
    streams) for fixed input shapes.  Inputs are copied into static buffers, the graph is replayed, the parameters get the graph's
    gradient tensors as ``.grad`` (the caller's optimiser then steps eagerly).  Per-site dropout seeds are frozen into the kernel
    arguments at capture; ``ag_set_dropout_salt`` before every replay makes each replay draw new keep patterns."""

    def __init__(self, trainer, xs, bits_loss, v_0, v_s, v_1, k, labels, train, seed):
        self.trainer = trainer
        self.static = [t.clone() if t is not None else None for t in (xs, bits_loss, v_0, v_s, v_1, labels)]
        self.k, self.train = k, train
        self.params = [p for p in trainer.m.parameters() if p.requires_grad]
        self.param_ids = tuple(id(p) for p in self.params)
        self.ptrs = tuple(p.data_ptr() for p in trainer.m.parameters())
        self.frozen_key = trainer.bank.keys(trainable=False)
        dev = xs.device
        trainer.bank.invalidate(trainable=True)          # the refresh of every trainable weight becomes part of the graph
        for lin in (trainer.cls, trainer.pool):
            if lin is not None:
                lin._wkey = None
        ops.set_dropout_salt(0, dev)
        for p in self.params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            sx, sb, s0, ss, s1, sl = self.static
            self.total, self.phi = trainer._loss_and_grads_eager(sx, sb, s0, ss, s1, k, sl, train, seed)
            self.parts = trainer.last_parts
        self.grads = [(p, p.grad) for p in self.params if p.grad is not None]
        for p in self.params:
            p.grad = None

    def valid(self, trainer) -> bool:
        return (self.ptrs == tuple(p.data_ptr() for p in trainer.m.parameters()) and self.frozen_key == trainer.bank.keys(trainable=False)
                and tuple(id(p) for p in trainer.m.parameters() if p.requires_grad) == self.param_ids)

    def run(self, xs, bits_loss, v_0, v_s, v_1, labels, salt: int):
        for dst, src in zip(self.static, (xs, bits_loss, v_0, v_s, v_1, labels)):
            if dst is not None:
                dst.copy_(src)
        ops.set_dropout_salt(salt, xs.device)
        self.graph.replay()
        ops.set_dropout_salt(0, xs.device)
        for p, g in self.grads:
            T._touch(p)
            p.grad = g
        self.trainer.last_parts = self.parts
        return self.total.clone(), self.phi


# ------------------------------------------------------------------------------------------------ trainers
class ExplainerTrainer16:
    """fw_explainer + loss_shapley_new with gradients (vanilla / froyo / duo; ViT or BERT) on the bf16 step."""

    def __init__(self, recipe, m_explainer: nn.Module):
        self.recipe, self.m = recipe, m_explainer
        cfg = m_explainer.config
        dev = next(m_explainer.parameters()).device
        self.is_vit = hasattr(m_explainer, "vit")
        self.duo = bool(recipe.training.exp_variant_duo) if recipe is not None else hasattr(m_explainer, "classifier")
        self.n_players = T._module_n_players(m_explainer)
        self.bank = WeightBank(dev)
        self.side = _Side.of(dev)
        self.backbone = ViTBackbone16(self.bank, m_explainer.vit) if self.is_vit else BertBackbone16(self.bank, m_explainer.bert)
        blk = ViTBlock16 if self.is_vit else BertBlock16
        self.attn = [blk(self.bank, ly, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob)
                     for ly in m_explainer.explainer_attn]
        self.mlp = MLPHead16(self.bank, m_explainer.explainer_mlp)
        # the duo heads work on B rows: the fp32-operand Linear of training.py
        self.cls = T.Lin([m_explainer.classifier]) if self.duo else None
        self.pool = T.Lin([m_explainer.bert_pooler.dense]) if (self.duo and not self.is_vit) else None
        self.step = 0
        self.saved = None
        self.use_graph = GRAPH_STEP
        self.graph_salt: Optional[int] = None     # tests: a fixed dropout salt for every replay (0 = the captured step's own patterns)
        self._graphs, self._seen = {}, set()

    def forward_phi(self, xs: Tensor, v_0: Optional[Tensor], v_1: Optional[Tensor], train: bool = True, seed: int = 0,
                    bits: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        cfg = self.m.config
        self.step += 1
        seeds = T.Seeds(seed * 7919 + self.step)
        b = xs.shape[0]
        p = self.n_players
        t, h, c = p + 1, cfg.hidden_size, cfg.num_labels
        if bits is None:
            bits = engine.ones_mask_bits(b, p, xs.device)
        self.bank.refresh()
        z, zb = self.backbone.forward(xs, bits, seeds, train)
        o, ob = z, zb
        n = len(self.attn)
        if self.is_vit:
            for i, blk in enumerate(self.attn):
                nxt = _ln(self.attn[i + 1].n1, cfg.layer_norm_eps) if i + 1 < n else self.mlp.next_ln()
                o, ob, _ = blk.forward(o, ob, bits, b, t, seeds, train, nxt, True)
            if n == 0 and self.mlp.ln is not None:
                _, _, ob = ops.rows_finish(o, ln=self.mlp.next_ln())
        else:
            for blk in self.attn:
                o, ob = blk.forward(o, ob, bits, b, t, seeds, train, True)
        s_exp = seeds.next()
        ph = cfg.hidden_dropout_prob if (train and not self.is_vit) else 0.0   # BERT explainer_dropout (models/vanilla_bert.py:152)
        if ph > 0.0:
            _, _, ob = ops.rows_finish(o, p_drop=ph, seed=s_exp)
        pred = self.mlp.forward(ob).view(b, t, c)
        phi = ops.shapley_normalize(pred, v_1, v_0, normalize=bool(cfg.explainer_normalize))
        base, duo_saved = None, None
        if self.duo:
            with _fp32_lin():
                zc = _cls_rows(z, b, t, h)
                if self.is_vit:
                    base = ops.softmax_rows(self.cls.forward(zc, L.AG_EPI_BIAS_F32))
                    duo_saved = (base,)
                else:
                    pooled = self.pool.forward(zc, L.AG_EPI_BIAS_TANH)
                    s_pool = seeds.next()
                    pd = cfg.hidden_dropout_prob if train else 0.0
                    base = self.cls.forward(ops.dropout(pooled, pd, s_pool), L.AG_EPI_BIAS_F32)
                    duo_saved = (pooled, pd, s_pool)
        self.saved = (b, t, h, c, ph, s_exp, duo_saved, o)
        return phi, base

    def backward_phi(self, dphi: Tensor, dbase: Optional[Tensor] = None) -> None:
        cfg = self.m.config
        b, t, h, c, ph, s_exp, duo_saved, o_last = self.saved
        self.saved = None
        side = self.side
        side.begin(self.use_graph or b * t < SIDE_MIN_ROWS)
        dev = dphi.device
        dz_extra = None
        if self.duo and dbase is not None:
            with _fp32_lin():
                if self.is_vit:
                    (probs,) = duo_saved
                    dz_cls = self.cls.backward(ops.softmax_rows_bwd(probs, dbase.contiguous().float()))
                else:
                    pooled, pd, s_pool = duo_saved
                    dp = ops.dropout(self.cls.backward(dbase.contiguous().float()), pd, s_pool)
                    dz_cls = self.pool.backward(ops.tanh_bwd(pooled, dp))
            dz_extra = torch.zeros((b, t, h), dtype=torch.float32, device=dev)
            dz_extra[:, 0, :].copy_(dz_cls)
            dz_extra = dz_extra.view(b * t, h)
        elif self.duo:
            self.cls.x = self.cls.xt = None
            if self.pool is not None:
                self.pool.x = self.pool.xt = None
        dpred = ops.shapley_normalize_bwd(dphi, t, normalize=bool(cfg.explainer_normalize)).view(b * t, c)
        dsl = self.mlp.backward(side, dpred)
        bb_frozen = self.backbone.frozen
        if self.is_vit:
            n = len(self.attn)
            if n > 0:
                top = self.attn[-1]
                dgh, dbh = _ln_grads(self.mlp.ln, dev) if self.mlp.ln is not None else (None, None)
                db2 = torch.empty(h, dtype=torch.float32, device=dev)
                d, dyb = ops.rows_ln_bwd(dsl, x=o_last if self.mlp.ln is not None else None,
                                         gamma=self.mlp.ln.weight.detach() if self.mlp.ln is not None else None,
                                         eps=self.mlp.ln.eps if self.mlp.ln is not None else 0.0, want_bf16=True, p_drop=top.ph, seed=top.s_f,
                                         dgamma=dgh, dbeta=dbh, dbias=db2)
                if self.mlp.ln is not None:
                    _give_ln_grads(side, self.mlp.ln, dgh, dbh)
                for i in reversed(range(n)):
                    below = _Below(self.attn[i - 1].ph, self.attn[i - 1].s_f, True) if i > 0 else NO_BELOW
                    res = self.attn[i].backward(side, d, dyb, db2, below, need_dx=not (i == 0 and bb_frozen and dz_extra is None))
                    if res is None:
                        d = None
                        break
                    d, dyb, db2 = res
            else:
                dgh, dbh = _ln_grads(self.mlp.ln, dev) if self.mlp.ln is not None else (None, None)
                d, _ = ops.rows_ln_bwd(dsl, x=o_last if self.mlp.ln is not None else None,
                                       gamma=self.mlp.ln.weight.detach() if self.mlp.ln is not None else None,
                                       eps=self.mlp.ln.eps if self.mlp.ln is not None else 0.0, dgamma=dgh, dbeta=dbh)
                if self.mlp.ln is not None:
                    _give_ln_grads(side, self.mlp.ln, dgh, dbh)
            if not bb_frozen:
                self.backbone.backward(side, d, dz_extra)
        else:
            dy, dy_add = dsl, None
            if ph > 0.0:
                d, _ = ops.rows_ln_bwd(dsl)
                dy = ops.dropout(d, ph, s_exp)
            for i in reversed(range(len(self.attn))):
                res = self.attn[i].backward(side, dy, dy_add, need_dx=not (i == 0 and bb_frozen and dz_extra is None))
                if res is None:
                    dy = None
                    break
                dy, dy_add = res
            if not bb_frozen:
                if dz_extra is not None:
                    dy_add = dz_extra if dy_add is None else ops.add(dy_add, dz_extra)
                self.backbone.backward(side, dy, dy_add)
        side.join()

    def loss_and_grads(self, xs: Tensor, bits_loss: Tensor, v_0: Tensor, v_s: Tensor, v_1: Tensor, n_mask_samples: int,
                       labels: Optional[Tensor] = None, train: bool = True, seed: int = 0):
        """One reference training-step body (scripts/train_explainer.py:182-196 / train_duo_explainer.py:180-196) -> (loss, phi).
        With ``use_graph`` (AG_TRAIN_GRAPH=1) the step is captured into a hipGraph at its second call with a given set of shapes
        and replayed from then on (gradients bit-identical to the eager step; phi is then the graph's static output buffer)."""
        if self.use_graph and T.GRAD_SINK is None and all(p.grad is None for p in self.m.parameters() if p.requires_grad):
            key = (tuple(xs.shape), xs.dtype, tuple(bits_loss.shape), tuple(v_s.shape), n_mask_samples, bool(train), labels is None)
            g = self._graphs.get(key)
            if g is not None and not g.valid(self):
                g = None
            if g is None and key in self._seen:
                g = self._graphs[key] = _StepGraph(self, xs, bits_loss, v_0, v_s, v_1, n_mask_samples, labels, train, seed)
            if g is not None:
                self.step += 1
                salt = ((seed * 7919 + self.step) * 2654435761) & 0xFFFFFFFF if self.graph_salt is None else self.graph_salt
                return g.run(xs, bits_loss, v_0, v_s, v_1, labels, salt=salt)
            self._seen.add(key)          # first call with these shapes: eager (allocates every lazily created buffer)
        return self._loss_and_grads_eager(xs, bits_loss, v_0, v_s, v_1, n_mask_samples, labels, train, seed)

    def _loss_and_grads_eager(self, xs: Tensor, bits_loss: Tensor, v_0: Tensor, v_s: Tensor, v_1: Tensor, n_mask_samples: int,
                              labels: Optional[Tensor] = None, train: bool = True, seed: int = 0):
        b = xs.shape[0]
        phi, base = self.forward_phi(xs, v_0, v_1, train, seed)
        loss, dphi = ops.shapley_loss(bits_loss, v_0, v_s, phi, b, n_mask_samples, want_grad=True)
        total, dbase = loss, None
        self.last_parts = (loss, None, base)
        if self.duo:
            ce, dbase = T._cross_entropy(base, labels)
            total = loss + ce
            self.last_parts = (loss, ce, base)
        self.backward_phi(dphi, dbase)
        return total, phi


class SurrogateTrainer16:
    """fw_surrogate on masked inputs + loss_logits_kl_divergence with gradients (scripts/train_surrogate.py:133-147)."""

    def __init__(self, recipe, m_surrogate: nn.Module):
        self.recipe, self.m = recipe, m_surrogate
        dev = next(m_surrogate.parameters()).device
        self.is_vit = hasattr(m_surrogate, "vit")
        self.n_players = T._module_n_players(m_surrogate)
        self.bank = WeightBank(dev)
        self.side = _Side.of(dev)
        self.backbone = ViTBackbone16(self.bank, m_surrogate.vit) if self.is_vit else BertBackbone16(self.bank, m_surrogate.bert)
        self.cls = T.Lin([m_surrogate.classifier])
        self.pool = None if self.is_vit else T.Lin([m_surrogate.bert_pooler.dense])
        self.step = 0
        self.saved = None

    def forward_probs(self, xs: Tensor, bits: Tensor, train: bool = True, seed: int = 0) -> Tensor:
        cfg = self.m.config
        self.step += 1
        seeds = T.Seeds(seed * 104729 + self.step)
        b = xs.shape[0]
        t, h = self.n_players + 1, cfg.hidden_size
        self.bank.refresh()
        z, _ = self.backbone.forward(xs, bits, seeds, train)
        zc = _cls_rows(z, b, t, h)
        pooled, ph, s_pool = None, 0.0, 0
        with _fp32_lin():
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
        self.side.begin(b * t < SIDE_MIN_ROWS)
        dlogits = ops.softmax_rows_bwd(probs, dprobs.contiguous().float())
        with _fp32_lin():
            if self.is_vit:
                dzc = self.cls.backward(dlogits)
            else:
                dp = ops.dropout(self.cls.backward(dlogits), ph, s_pool)
                dzc = self.pool.backward(ops.tanh_bwd(pooled, dp))
        dz = torch.zeros((b, t, h), dtype=torch.float32, device=dprobs.device)
        dz[:, 0, :].copy_(dzc)
        if not self.backbone.frozen:
            self.backbone.backward(self.side, dz.view(b * t, h))
        self.side.join()

    def loss_and_grads(self, xs: Tensor, bits: Tensor, orig_probs: Tensor, train: bool = True, seed: int = 0):
        probs = self.forward_probs(xs, bits, train, seed)
        loss, dprobs = ops.kl_loss(orig_probs, probs, want_grad=True)
        self.backward_probs(dprobs)
        return loss, probs


class _fp32_lin:
    """the B-row duo / surrogate heads run training.Lin in its exact-fp32 form (their GEMMs are a few rows: nothing to gain from
    bf16 operands, and training.Lin's bf16 branch would cast and transpose them)."""

    def __enter__(self):
        self.keep = T.MIXED_BF16
        T.MIXED_BF16 = False

    def __exit__(self, *exc):
        T.MIXED_BF16 = self.keep
        return False
