"""Host-side driver of the masked forward: packs module parameters into the layout the HIP
kernels want and strings the C-ABI calls together for the recipes' ``fw_*`` callables.

Nothing here computes: tensors are allocated by torch, every op is a libautognothi_hip kernel.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _lib as L
from . import ops

_PRECISION = {"dtype": L.AG_BF16}
PRUNE_BERT_TOKENS = os.environ.get("AG_BERT_PRUNE", "1") != "0"  # BERT cls-only forwards skip additively masked tokens after layer 0
_PACKED_ROWS = {}      # device -> int32 [1]: visible tokens of the last pruned forward (stays on the device: no host read per step)


def last_packed_rows(device=None) -> int:
    """visible tokens of the last token-pruned BERT forward on `device` (bench.py: executed-work accounting).  Synchronises:
    call it outside timed regions."""
    if not _PACKED_ROWS:
        return 0
    key = str(device) if device is not None else next(iter(_PACKED_ROWS))
    t = _PACKED_ROWS.get(key)
    if t is None:
        return 0
    return int(t) if isinstance(t, int) else int(t.item())


def note_packed_rows(device, n) -> None:
    """(the LTT-BERT ladder driver: a host int, or the device int32 [1] tensor that holds the count)"""
    _PACKED_ROWS[str(device)] = n if isinstance(n, torch.Tensor) else int(n)
FOLD_LAYERNORM = os.environ.get("AG_LN_FOLD", "1") != "0"  # bf16 ViT: fold LayerNorm into the consuming GEMM epilogue


def set_precision(name: str) -> None:
    """'bf16' (throughput mode, BASELINE config 2) or 'fp32' (exact-fp32 MFMA; the mode in which the
    1e-4 Shapley-value parity criterion is checked)."""
    _PRECISION["dtype"] = {"bf16": L.AG_BF16, "fp32": L.AG_F32, "f32": L.AG_F32}[name]


def get_precision() -> int:
    return _PRECISION["dtype"]


def precision_name() -> str:
    return "bf16" if _PRECISION["dtype"] == L.AG_BF16 else "fp32"


class _Workspace:
    """One growing HBM scratch buffer per device (sized for 288 GB parts: never shrinks)."""

    def __init__(self):
        self.buf: Dict[str, Tensor] = {}

    def get(self, device: torch.device, nbytes: int) -> Tensor:
        key = str(device)
        cur = self.buf.get(key)
        if cur is None or cur.numel() < nbytes:
            self.buf[key] = torch.empty(int(nbytes * 1.05) + 256, dtype=torch.uint8, device=device)
        return self.buf[key]


WORKSPACE = _Workspace()


def param_key(p: Tensor) -> Tuple:
    """identity of a parameter's current VALUE for the weight caches: storage, torch's in-place version counter, and the count
    of gradients this package's backward has written into it (``_ag_step``, bumped by training._acc_grad / _grad).  The last
    one matters: ``torch.optim.AdamW(fused=True)`` (and the other fused / foreach optimisers that update through
    ``torch._fused_*``) do NOT advance ``_version``, so a cache keyed on it alone keeps serving the weights of step 1."""
    return (p.data_ptr(), p._version, p.__dict__.get("_ag_step", 0), _WEIGHTS_EPOCH[0])


_WEIGHTS_EPOCH = [0]


def invalidate_weight_caches() -> None:
    """Rebuild every weight cache before its next use.  Needed only after parameter updates nothing above can see: in-place
    writes through ``p.data`` (which carries no version counter) by code that never ran this package's backward."""
    _WEIGHTS_EPOCH[0] += 1


def watch_optimizer(optimizer) -> None:
    """Make every ``optimizer.step()`` visible to the weight caches: a post-step hook bumps ``_ag_step`` of exactly the
    parameters the optimiser owns (fused / foreach optimisers update without advancing ``Tensor._version``, and a forward
    between ``backward`` and ``step`` would otherwise refill the caches under the post-backward key and keep serving the
    pre-step weights).  Idempotent; the train_* entry points call it on the optimisers they build, the epoch bodies on the
    one they are handed."""
    if getattr(optimizer, "_ag_watched", False):
        return

    def bump(opt, *args, **kwargs):
        # only what the step UPDATED: torch's optimisers skip parameters whose .grad is None, and the entry points hand
        # `m_explainer.parameters()` — frozen froyo / LTT backbones included — to AdamW; bumping those would rebuild every pack
        # of the frozen weights (and re-capture every GraphedStep) after each step
        for group in opt.param_groups:
            for q in group["params"]:
                if q.grad is not None:
                    q.__dict__["_ag_step"] = q.__dict__.get("_ag_step", 0) + 1
    optimizer.register_step_post_hook(bump)
    optimizer._ag_watched = True


_CAPTURE_LOG: Optional[list] = None     # GraphedStep: the packs a capture baked pointers of


def _log_pack(pack, key, tensors) -> None:
    if _CAPTURE_LOG is not None:
        _CAPTURE_LOG.append((pack, key, tensors))


def _versions(params: Sequence[Tensor]) -> Tuple:
    return tuple(param_key(p) for p in params)


class PackedLinear:
    """weight [N,K] in the storage dtype + fp32 bias, rebuilt when the parameters change."""

    def __init__(self, weights: Sequence[Tensor], biases: Sequence[Tensor]):
        self.weights, self.biases = list(weights), list(biases)
        self.key = None
        self.w: Optional[Tensor] = None
        self.b: Optional[Tensor] = None

    def get(self, dtype: int) -> Tuple[Tensor, Tensor]:
        key = (dtype, _versions(self.weights + self.biases))
        if key != self.key:
            with torch.no_grad():
                w = torch.cat([x.detach().reshape(x.shape[0], -1) for x in self.weights], dim=0).float().contiguous()
                self.w = ops.cast(w, dtype)
                self.b = torch.cat([x.detach().float() for x in self.biases], dim=0).contiguous()
            self.key = key
        _log_pack(self, key, (self.w, self.b))
        return self.w, self.b

    def current_key(self, dtype: int) -> Tuple:
        return (dtype, _versions(self.weights + self.biases))


class PackedFoldedLinear:
    """Linear(LayerNorm(x)) folded for the GEMM epilogue (ViT pre-LN, bf16): W' = gamma ⊙ W (storage dtype),
    bias' = b + W·beta, colsum[n] = sum_k W'[n,k] (of the rounded W').  Pack-time preprocessing, rebuilt when
    any of the parameters changes."""

    def __init__(self, weights: Sequence[Tensor], biases: Sequence[Tensor], ln: nn.Module):
        self.weights, self.biases, self.ln = list(weights), list(biases), ln
        self.key = None
        self.w = self.b = self.s = None

    def get(self, dtype: int) -> Tuple[Tensor, Tensor, Tensor]:
        key = (dtype, _versions(self.weights + self.biases + [self.ln.weight, self.ln.bias]))
        if key != self.key:
            with torch.no_grad():   # one launch per weight (q | k | v side by side): gamma * W rounded, b + W.beta, column sums of the rounded W'
                self.w, self.b, self.s = ops.pack_folded_linear(self.weights, self.biases, self.ln.weight, self.ln.bias, dtype)
            self.key = key
        _log_pack(self, key, (self.w, self.b, self.s))
        return self.w, self.b, self.s

    def current_key(self, dtype: int) -> Tuple:
        return (dtype, _versions(self.weights + self.biases + [self.ln.weight, self.ln.bias]))


def _f32(p: Optional[Tensor]) -> Optional[Tensor]:
    return None if p is None else p.detach().float().contiguous()


class PackedEncoder:
    """A stack of transformer layers (reference VanillaViTLayer / VanillaBertLayer modules)."""

    def __init__(self, layers: Sequence[nn.Module], kind: int, T: int, H: int, I: int, heads: int, eps: float):
        self.kind, self.T, self.H, self.I, self.heads, self.eps = kind, T, H, I, heads, eps
        self.layers = list(layers)
        self.lin = []
        for ly in self.layers:
            att = ly.attention
            self.lin.append(dict(
                qkv=PackedLinear([att.self.query.weight, att.self.key.weight, att.self.value.weight],
                                 [att.self.query.bias, att.self.key.bias, att.self.value.bias]),
                o=PackedLinear([att.output.dense.weight], [att.output.dense.bias]),
                fc1=PackedLinear([ly.intermediate.dense.weight], [ly.intermediate.dense.bias]),
                fc2=PackedLinear([ly.output.dense.weight], [ly.output.dense.bias]),
            ))
            if kind == L.AG_MASK_VIT_MUL:  # LN-folded variants of the two projections that consume a LayerNorm
                if not isinstance(ly.layernorm_before, nn.Identity):
                    self.lin[-1]["qkv_ln"] = PackedFoldedLinear([att.self.query.weight, att.self.key.weight, att.self.value.weight],
                                                                [att.self.query.bias, att.self.key.bias, att.self.value.bias],
                                                                ly.layernorm_before)
                self.lin[-1]["fc1_ln"] = PackedFoldedLinear([ly.intermediate.dense.weight], [ly.intermediate.dense.bias],
                                                            ly.layernorm_after)
            else:   # BERT (post-LN): QKV consumes the PREVIOUS layer's output.LayerNorm, fc1 this layer's attention.output.LayerNorm
                i_ly = len(self.lin) - 1
                if i_ly >= 1:
                    self.lin[-1]["qkv_ln"] = PackedFoldedLinear([att.self.query.weight, att.self.key.weight, att.self.value.weight],
                                                                [att.self.query.bias, att.self.key.bias, att.self.value.bias],
                                                                self.layers[i_ly - 1].output.LayerNorm)
                if not isinstance(att.output.LayerNorm, nn.Identity):
                    self.lin[-1]["fc1_ln"] = PackedFoldedLinear([ly.intermediate.dense.weight], [ly.intermediate.dense.bias],
                                                                att.output.LayerNorm)
        self._keep: List = []

    def _ln(self, ly: nn.Module, which: int) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        if self.kind == L.AG_MASK_VIT_MUL:
            mod = ly.layernorm_before if which == 1 else ly.layernorm_after
        else:
            mod = ly.attention.output.LayerNorm if which == 1 else ly.output.LayerNorm
        if isinstance(mod, nn.Identity):
            return None, None
        return _f32(mod.weight), _f32(mod.bias)

    def desc(self, dtype: int) -> L.ag_encoder_desc:
        arr = (L.ag_layer_weights * len(self.layers))()
        keep = []
        for i, ly in enumerate(self.layers):
            lw = arr[i]
            for name, cw, cb in (("qkv", "w_qkv", "b_qkv"), ("o", "w_o", "b_o"), ("fc1", "w_fc1", "b_fc1"), ("fc2", "w_fc2", "b_fc2")):
                w, b = self.lin[i][name].get(dtype)
                keep += [w, b]
                setattr(lw, cw, w.data_ptr())
                setattr(lw, cb, b.data_ptr())
            # (BERT: only the token-pruned forwards — ag_bert_encoder_forward_pruned, ag_bert_layers_forward_packed — read the folded forms)
            if dtype == L.AG_BF16 and FOLD_LAYERNORM and (self.kind == L.AG_MASK_VIT_MUL or PRUNE_BERT_TOKENS):
                for name, cw, cb, cs in (("qkv_ln", "w_qkv_ln", "b_qkv_ln", "s_qkv_ln"), ("fc1_ln", "w_fc1_ln", "b_fc1_ln", "s_fc1_ln")):
                    if name in self.lin[i]:
                        w, b, s_ = self.lin[i][name].get(dtype)
                        keep += [w, b, s_]
                        setattr(lw, cw, w.data_ptr()); setattr(lw, cb, b.data_ptr()); setattr(lw, cs, s_.data_ptr())
            g1, b1 = self._ln(ly, 1)
            g2, b2 = self._ln(ly, 2)
            keep += [g1, b1, g2, b2]
            lw.ln1_g, lw.ln1_b = L.ptr(g1), L.ptr(b1)
            lw.ln2_g, lw.ln2_b = L.ptr(g2), L.ptr(b2)
        d = L.ag_encoder_desc(self.kind, dtype, self.T, self.H, self.I, self.heads, self.eps, len(self.layers), arr)
        self._keep = [keep, arr]
        return d

    def forward(self, h0: Tensor, rows: int, share: int, mask_bits: Tensor, cls_only_last: bool, dtype: int,
                chain: Optional[list] = None) -> Tensor:
        """h0 [rows/share, T, H] -> hidden [rows, T, H], both in the storage dtype (the residual stream).
        ``chain`` = [row_stats fp32 [rows*T*2], stats_ready, want_stats_out]: layer-by-layer callers hand the LayerNorm-fold
        row statistics from one call to the next (ag_encoder_forward_chained); chain[1] is updated in place with whether
        this call left valid statistics of its output behind."""
        L.require_gpu(h0, mask_bits)
        if h0.dtype != ops.storage_dtype(dtype):
            h0 = ops.cast(h0, dtype)
        h0 = h0.contiguous()
        d = self.desc(dtype)
        out = torch.empty((rows, self.T, self.H), dtype=ops.storage_dtype(dtype), device=h0.device)
        with L.on(h0.device):
            need = L.lib().ag_encoder_workspace_bytes(C.byref(d), rows)
            ws = WORKSPACE.get(h0.device, need)
            if self.kind == L.AG_MASK_BERT_ADD and cls_only_last and len(self.layers) >= 2 and PRUNE_BERT_TOKENS:
                key = str(h0.device)
                if not isinstance(_PACKED_ROWS.get(key), torch.Tensor):
                    _PACKED_ROWS[key] = torch.zeros(1, dtype=torch.int32, device=h0.device)
                L.check(L.lib().ag_bert_encoder_forward_pruned(C.byref(d), L.ptr(h0), rows, share, L.ptr(mask_bits), L.ptr(out),
                                                               L.ptr(ws), ws.numel(), L.ptr(_PACKED_ROWS[key]), L.stream()))
                return out
            if chain is not None:
                st, st_in, st_out = chain
                written = C.c_int32(0)
                L.check(L.lib().ag_encoder_forward_chained(C.byref(d), L.ptr(h0), rows, share, L.ptr(mask_bits), L.ptr(out),
                                                           1 if cls_only_last else 0, L.ptr(ws), ws.numel(), L.ptr(st),
                                                           1 if st_in else 0, 1 if st_out else 0, C.byref(written), L.stream()))
                chain[1] = bool(written.value)
                return out
            L.check(L.lib().ag_encoder_forward(C.byref(d), L.ptr(h0), rows, share, L.ptr(mask_bits), L.ptr(out),
                                               1 if cls_only_last else 0, L.ptr(ws), ws.numel(), L.stream()))
        return out


def _forward_packed(self, x: Tensor, cu: Tensor, rows: int, n_packed: int, dtype: int, rows_dev: Optional[Tensor] = None) -> Tensor:
    """(BERT kind) this encoder's layers on packed rows x [N, H] (visible tokens of `rows` sequences, cu_seqlens) -> [N, H].
    rows_dev: device int32 [1] holding the actual packed row count (n_packed is then the upper bound)."""
    L.require_gpu(x, cu)
    if self.kind != L.AG_MASK_BERT_ADD:
        raise ValueError("packed (token-pruned) layers exist for the additive BERT mask only")
    x = x.contiguous()
    d = self.desc(dtype)
    out = torch.empty((n_packed, self.H), dtype=ops.storage_dtype(dtype), device=x.device)
    with L.on(x.device):
        need = L.lib().ag_encoder_workspace_bytes(C.byref(d), rows)
        ws = WORKSPACE.get(x.device, need)
        L.check(L.lib().ag_bert_layers_forward_packed(C.byref(d), L.ptr(x), L.ptr(cu), rows, n_packed, L.ptr(out), L.ptr(ws),
                                                      ws.numel(), L.ptr(rows_dev), L.stream()))
    return out


PackedEncoder.forward_packed = _forward_packed


_ONES_BITS = {}


def ones_mask_bits(rows: int, n_players: int, device: torch.device) -> Tensor:
    """Key bits of an all-ones mask [rows, P] with CLS prepended (bits beyond T are zero).  Cached per shape: building it
    is a pageable host-to-device copy, i.e. a stream synchronisation in the middle of every training step."""
    key = (rows, n_players, str(device))
    if key not in _ONES_BITS:
        if len(_ONES_BITS) > 64:
            _ONES_BITS.clear()
        _ONES_BITS[key] = _ones_mask_bits(rows, n_players, device)
    return _ONES_BITS[key]


def _ones_mask_bits(rows: int, n_players: int, device: torch.device) -> Tensor:
    t = n_players + 1
    tw = ops.mask_words(n_players)
    words = []
    for w in range(tw):
        nbits = max(0, min(32, t - 32 * w))
        v = (1 << nbits) - 1
        words.append(v - (1 << 32) if v >= (1 << 31) else v)
    return torch.tensor(words, dtype=torch.int32, device=device).repeat(rows, 1).contiguous()


def to_mask_bits(attention_mask: Tensor, n_players: int) -> Tensor:
    """Accept what the reference passes to ``forward`` — an int64 [R, T] mask whose first column is the
    CLS column (recipes/*._fw_xs_preprocess) — or pre-packed int32 key bits [R, Tw]."""
    if attention_mask.dtype == torch.int32 and attention_mask.shape[1] == ops.mask_words(n_players):
        return attention_mask.contiguous()
    if attention_mask.shape[1] != n_players + 1:
        raise ValueError(f"attention_mask must be [R, {n_players + 1}] (CLS + players), got {tuple(attention_mask.shape)}")
    # the CLS column is defined to be 1 by every caller; the packed form hard-wires it
    return ops.pack_mask(attention_mask[:, 1:])


def linear_head(x_rows: Tensor, lda: int, m: int, lin: PackedLinear, epilogue: int, dtype: int) -> Tensor:
    w, b = lin.get(dtype)
    return ops.gemm(x_rows, w, b, epilogue, dtype, m=m, lda=lda)


class GraphedStep:
    """One step of the hot path as a hipGraph: ``fn`` (no arguments; reads static device tensors, e.g. the resident inputs,
    and returns device tensors) is captured once and replayed with a single ``hipGraphLaunch``.

    At the reference's operating point (2-4 inputs x K=32 masks per GPU, experiments/*/.hparams.json) a forward is ~170 kernel
    launches of a few microseconds each: host launch overhead, not the GPU, sets the pace.  Everything in the step is
    capture-safe by construction: the mask sampler's generator state lives in HBM (each replay advances it exactly as an eager
    call would), the encoder is one C call with no allocation or synchronisation, LayerNorm-fold statistics need no zero fill.
    Capture goes through torch.cuda.CUDAGraph (= hipStreamBeginCapture / hipGraphInstantiate on the launch stream) so that
    the tensors ``fn`` allocates come from a pool owned by the graph and stay valid across replays.
    The token-pruned BERT forward is capturable too: its data-dependent packed row count stays on the device
    (the d_rows arguments of the C ABI).  Not capturable: host reads (``.item()``), in-library event timing (ag_profile_enable).
    The graph keeps the workspace buffer and the weight packs of its capture alive and re-captures itself when a parameter
    behind one of them has changed (``check=False`` skips that test: ~50 us of host time per replay on ViT-base)."""

    def __init__(self, fn, warmup: int = 2, check: bool = True):
        self.fn, self.warmup, self.check = fn, warmup, check
        self._capture()

    def _capture(self) -> None:
        global _CAPTURE_LOG
        fn = self.fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # weight packing, workspace growth, lazy kernel attributes: all before capture
            for _ in range(max(1, self.warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        _CAPTURE_LOG = []
        try:
            with torch.cuda.graph(self.graph):
                self.out = fn()
            packs = _CAPTURE_LOG
        finally:
            _CAPTURE_LOG = None
        # The graph holds RAW device pointers of storage it does not own: the shared workspace and the packed weights.  Keep
        # references to exactly the tensors that were live at capture (a later, larger forward may make WORKSPACE drop its
        # buffer; an optimiser step rebuilds the packs) so that a replay can never write into memory the allocator has handed
        # out again, and remember what they were packed FROM so that a replay on stale weights is detected.
        self._ws = dict(WORKSPACE.buf)
        seen, self._packs = set(), []
        for pack, key, tensors in packs:
            if id(pack) not in seen:
                seen.add(id(pack))
                self._packs.append((pack, key, tensors))

    def stale(self) -> bool:
        """have the weights a captured launch reads been updated since capture?"""
        return any(pack.current_key(key[0]) != key for pack, key, _ in self._packs)

    def __call__(self):
        if self.check and self.stale():        # parameters changed (optimizer.step(), load_state_dict): capture again on the new packs
            self._capture()
        self.graph.replay()
        return self.out
