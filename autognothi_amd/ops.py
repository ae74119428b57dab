"""Tensor-level wrappers over the C ABI (one Python function per entry point).

torch is plumbing here: it owns the HBM allocations and the HIP stream; every computation is a
kernel of ``libautognothi_hip.so`` launched on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import ctypes as C

import numpy as np
import torch
from torch import Tensor

from . import _lib as L

F32, BF16 = L.AG_F32, L.AG_BF16


def storage_dtype(dtype: int) -> torch.dtype:
    return torch.bfloat16 if dtype == BF16 else torch.float32


def mask_words(n_players: int) -> int:
    return (n_players + 1 + 31) // 32


# ----------------------------------------------------------------------------- RNG / samplers
def parse_torch_cpu_state(state: Tensor):
    """torch.get_rng_state() of the CPU generator -> (mt[624] uint32, pos).  Layout (at::mt19937_data_pod): u64 seed,
    i32 left, i32 seeded, u64 next, u64 state[624], ...; at::mt19937 keeps ``left == 625 - next`` between twists and
    ``left == 1`` when the next draw twists first (fresh seed: next == 0; exhausted block: next == 624).
    pos: index of the next word to hand out, 624 = twist on the next draw."""
    st = state.numpy()
    left = int(st[8:12].view(np.int32)[0])
    nxt = int(st[16:24].view(np.uint64)[0])
    mt = np.ascontiguousarray(st[24:24 + 624 * 8].view(np.uint64).astype(np.uint32))
    return mt, (624 if left == 1 else nxt)


def fill_torch_cpu_state(state: Tensor, mt: np.ndarray, pos: int) -> Tensor:
    """Inverse of parse_torch_cpu_state on a copy of `state` (Box-Muller cache fields are left as they are)."""
    st = state.clone()
    a = st.numpy()
    a[8:12].view(np.int32)[0] = 1 if pos >= 624 else 625 - pos
    a[12:16].view(np.int32)[0] = 1
    a[16:24].view(np.uint64)[0] = 624 if pos >= 624 else pos
    a[24:24 + 624 * 8].view(np.uint64)[:] = np.asarray(mt, dtype=np.uint32).astype(np.uint64)
    return st


class DeviceMT19937:
    """MT19937 state resident in HBM; bit-compatible with torch's CPU generator
    (reference consumers: models/shapley.py:69,114,133)."""

    def __init__(self, device: torch.device, seed: Optional[int] = None):
        self.device = torch.device(device)
        self.state = torch.zeros(L.AG_MT_STATE_BYTES // 4, dtype=torch.int32, device=self.device)
        if seed is not None:
            self.seed(seed)

    def seed(self, seed: int) -> "DeviceMT19937":
        with L.on(self.device):
            L.check(L.lib().ag_mt19937_seed(L.ptr(self.state), seed & 0xFFFFFFFF, L.stream()))
        return self

    def import_torch_cpu_state(self, gen: Optional[torch.Generator] = None) -> "DeviceMT19937":
        """Continue torch's *CPU* generator stream on the device."""
        mt, pos = parse_torch_cpu_state(gen.get_state() if gen is not None else torch.get_rng_state())
        with L.on(self.device):
            L.check(L.lib().ag_mt19937_import(L.ptr(self.state), mt.ctypes.data, pos, L.stream()))
        return self

    def export_to_torch_cpu(self, gen: Optional[torch.Generator] = None) -> None:
        """Write the advanced state back into torch's CPU generator (synchronises)."""
        mt = np.zeros(624, dtype=np.uint32)
        import ctypes as C
        pos = C.c_int(0)
        with L.on(self.device):
            L.check(L.lib().ag_mt19937_export(L.ptr(self.state), mt.ctypes.data, C.byref(pos), L.stream()))
        st = fill_torch_cpu_state(gen.get_state() if gen is not None else torch.get_rng_state(), mt, pos.value)
        (gen.set_state(st) if gen is not None else torch.set_rng_state(st))

    def skip(self, n: int) -> "DeviceMT19937":
        """advance by n draws without producing them."""
        if n > 0:
            with L.on(self.device):
                L.check(L.lib().ag_mt19937_skip(L.ptr(self.state), int(n), L.stream()))
        return self

    def raw(self, n: int) -> Tensor:
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        with L.on(self.device):
            L.check(L.lib().ag_mt19937_raw(L.ptr(self.state), L.ptr(out), n, L.stream()))
        return out


_PREFIX_CACHE = {}


def shapley_prefix_table(n_players: int, device: torch.device) -> Tensor:
    """The size-prior prefix table of reference models/shapley.py:65-67,:132, computed once per P
    with the same torch CPU ops the reference uses (sum / cumsum reduction order is part of the
    bit-exact contract) and kept resident on the device."""
    key = (n_players, str(device))
    if key not in _PREFIX_CACHE:
        k = torch.arange(1, n_players)
        probs = 1 / (k * (n_players - k))
        probs = probs / probs.sum()
        prefix = torch.cumsum(probs, dim=0) - probs
        _PREFIX_CACHE[key] = prefix.to(torch.float32).to(device)
    return _PREFIX_CACHE[key]


def mask_shapley_new(rng: DeviceMT19937, n_mask_samples: int, n_players: int, want_i64: bool = True,
                     want_bits: bool = True, prefix: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """reference models/shapley.py:56-79 -> (int64 [n,P] masks, uint32-as-int32 [n,Tw] key bits)."""
    if n_mask_samples % 2 != 0:
        raise AssertionError("n_mask_samples must be even (reference models/shapley.py:62)")
    dev = rng.device
    prefix = shapley_prefix_table(n_players, dev) if prefix is None else prefix
    L.require_gpu(rng.state, prefix)
    mi = torch.empty((n_mask_samples, n_players), dtype=torch.int64, device=dev) if want_i64 else None
    mb = torch.empty((n_mask_samples, mask_words(n_players)), dtype=torch.int32, device=dev) if want_bits else None
    scratch = torch.empty(max(1, n_mask_samples // 2 * (n_players + 1)), dtype=torch.int32, device=dev)
    with L.on(dev):
        L.check(L.lib().ag_mask_shapley_new(L.ptr(rng.state), n_mask_samples, n_players, L.ptr(prefix),
                                            L.ptr(mi), L.ptr(mb), L.ptr(scratch), L.stream()))
    return mi, mb


def mask_shapley_new_rows(rng: DeviceMT19937, n_mask_samples_total: int, row_lo: int, row_hi: int, n_players: int,
                          want_i64: bool = True, want_bits: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """rows [row_lo, row_hi) of ``mask_shapley_new(n_mask_samples_total, n_players)``; the generator advances by the whole
    call (row-sharded ranks: autognothi_amd.distributed.ShardedMaskStream)."""
    dev = rng.device
    prefix = shapley_prefix_table(n_players, dev)
    n = row_hi - row_lo
    mi = torch.empty((n, n_players), dtype=torch.int64, device=dev) if want_i64 else None
    mb = torch.empty((n, mask_words(n_players)), dtype=torch.int32, device=dev) if want_bits else None
    scratch = torch.empty(max(1, n // 2 * (n_players + 1)), dtype=torch.int32, device=dev)
    with L.on(dev):
        L.check(L.lib().ag_mask_shapley_new_rows(L.ptr(rng.state), n_mask_samples_total, row_lo, row_hi, n_players, L.ptr(prefix),
                                                 L.ptr(mi), L.ptr(mb), L.ptr(scratch), L.stream()))
    return mi, mb


def mask_purely_uniform(rng: DeviceMT19937, batch: int, n_players: int, want_i64: bool = True,
                        want_bits: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """reference models/shapley.py:109-115."""
    dev = rng.device
    mi = torch.empty((batch, n_players), dtype=torch.int64, device=dev) if want_i64 else None
    mb = torch.empty((batch, mask_words(n_players)), dtype=torch.int32, device=dev) if want_bits else None
    scratch = torch.empty(max(1, batch * (n_players + 1)), dtype=torch.int32, device=dev)
    with L.on(dev):
        L.check(L.lib().ag_mask_purely_uniform(L.ptr(rng.state), batch, n_players, L.ptr(mi), L.ptr(mb),
                                               L.ptr(scratch), L.stream()))
    return mi, mb


def mask_purely_uniform_rows(rng: DeviceMT19937, batch_total: int, row_lo: int, row_hi: int, n_players: int, want_i64: bool = True,
                             want_bits: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """rows [row_lo, row_hi) of ``mask_purely_uniform(batch_total, n_players)``; the generator advances by the whole call."""
    dev = rng.device
    n = row_hi - row_lo
    mi = torch.empty((n, n_players), dtype=torch.int64, device=dev) if want_i64 else None
    mb = torch.empty((n, mask_words(n_players)), dtype=torch.int32, device=dev) if want_bits else None
    scratch = torch.empty(max(1, n * (n_players + 1)), dtype=torch.int32, device=dev)
    with L.on(dev):
        L.check(L.lib().ag_mask_purely_uniform_rows(L.ptr(rng.state), batch_total, row_lo, row_hi, n_players, L.ptr(mi), L.ptr(mb),
                                                    L.ptr(scratch), L.stream()))
    return mi, mb


def pack_mask(mask_i64: Tensor) -> Tensor:
    """[R,P] int64 0/1 -> [R, ceil((P+1)/32)] key bits with the always-on CLS bit prepended
    (recipes/vanilla_vit.py:219-224)."""
    L.require_gpu(mask_i64)
    m = mask_i64.contiguous()
    if m.dtype != torch.int64:
        m = m.to(torch.int64)
    rows, p = m.shape
    bits = torch.empty((rows, mask_words(p)), dtype=torch.int32, device=m.device)
    with L.on(m.device):
        L.check(L.lib().ag_pack_mask(L.ptr(m), rows, p, L.ptr(bits), L.stream()))
    return bits


def perturbed_masks(attr: Tensor, steps: int, mask_base: int) -> Tuple[Tensor, Tensor]:
    """scripts/measure_faithfulness.py:225-251 for attr [n_attr, P] -> (stops [S], masks [n_attr,S,P])."""
    L.require_gpu(attr)
    a = attr.contiguous().float()
    n_attr, p = a.shape
    s = min(p, steps)
    stops = torch.empty(s, dtype=torch.int64, device=a.device)
    masks = torch.empty((n_attr, s, p), dtype=torch.int64, device=a.device)
    with L.on(a.device):
        L.check(L.lib().ag_perturbed_masks(L.ptr(a), n_attr, p, steps, mask_base, L.ptr(stops), L.ptr(masks), L.stream()))
    return stops, masks


# ----------------------------------------------------------------------------- building blocks
def cast(src: Tensor, dtype: int) -> Tensor:
    L.require_gpu(src)
    s = src.contiguous().float()
    out = torch.empty(s.shape, dtype=storage_dtype(dtype), device=s.device)
    with L.on(s.device):
        L.check(L.lib().ag_cast_f32(L.ptr(s), L.ptr(out), s.numel(), dtype, L.stream()))
    return out


def pack_folded_linear(weights, biases, gamma: Tensor, beta: Tensor, dtype: int):
    """Linear(LayerNorm(x)) folded for the GEMM epilogue (ag_pack_folded_linear): the [N_i, K] fp32 weights (and [N_i] biases, entries
    may be None) land side by side -> (W' = gamma * W in the storage dtype [N, K], b' = b + W.beta [N], colsum of the rounded W' [N])."""
    L.require_gpu(gamma, beta, *weights)
    k = weights[0][0].numel()
    n_tot = sum(int(w_.shape[0]) for w_ in weights)
    dev = gamma.device
    w_out = torch.empty((n_tot, k), dtype=storage_dtype(dtype), device=dev)
    b_out = torch.empty(n_tot, dtype=torch.float32, device=dev)
    s_out = torch.empty(n_tot, dtype=torch.float32, device=dev)
    g, bt = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    n0 = 0
    with L.on(dev):
        for w_, b_ in zip(weights, biases):
            n = int(w_.shape[0])
            wf = w_.detach().reshape(n, -1).float().contiguous()
            bf = None if b_ is None else b_.detach().float().contiguous()
            L.check(L.lib().ag_pack_folded_linear(L.ptr(wf), L.ptr(bf), L.ptr(g), L.ptr(bt), n, k, L.ptr(w_out[n0:]), dtype,
                                                  L.ptr(b_out[n0:]), L.ptr(s_out[n0:]), L.stream()))
            n0 += n
    return w_out, b_out, s_out


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, dtype: int, rows: Optional[int] = None,
              ldx: Optional[int] = None, want_store: bool = True, want_f32: bool = False, rows_dev: Optional[Tensor] = None):
    """x [..., H] fp32 or bf16 (or a strided view described by rows/ldx); statistics in fp32.  ``rows_dev`` (here and in gemm /
    gemm_resid_ln / gather_rows / side_*): device int32 tensor whose first element is the ACTUAL row count — ``rows`` is then
    an upper bound that sizes the launch (the C ABI's ``d_rows`` argument)."""
    L.require_gpu(x, gamma, beta)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"layernorm: unsupported input dtype {x.dtype}")
    x_dtype = F32 if x.dtype == torch.float32 else BF16
    h = gamma.numel()
    rows = x.numel() // h if rows is None else rows
    ldx = h if ldx is None else ldx
    ys = torch.empty((rows, h), dtype=storage_dtype(dtype), device=x.device) if want_store else None
    yf = torch.empty((rows, h), dtype=torch.float32, device=x.device) if want_f32 else None
    with L.on(x.device):
        L.check(L.lib().ag_layernorm(L.ptr(x), x_dtype, ldx, rows, h, L.ptr(gamma), L.ptr(beta), eps, L.ptr(ys), L.ptr(yf),
                                     dtype, L.ptr(rows_dev), L.stream()))
    return ys, yf


def gemm(a: Tensor, w: Tensor, bias: Optional[Tensor], epilogue: int, dtype: int, m: Optional[int] = None,
         lda: Optional[int] = None, resid: Optional[Tensor] = None, ldr: Optional[int] = None,
         rows_per_seq: int = 1, resid_share: int = 1, out: Optional[Tensor] = None, ldc: Optional[int] = None,
         ln_stats: Optional[Tensor] = None, ln_colsum: Optional[Tensor] = None, ln_eps: float = 0.0,
         stats_out: Optional[Tensor] = None, rows_dev: Optional[Tensor] = None) -> Tensor:
    """epilogue(A[M,K] @ W[N,K]^T + bias).  a, w, resid in the storage dtype; bias fp32.  Output: fp32 for
    AG_EPI_BIAS_F32, else the storage dtype."""
    L.require_gpu(a, w, bias, resid, out)
    n, k = w.shape
    m = a.numel() // k if m is None else m
    lda = k if lda is None else lda
    out_f32 = epilogue == L.AG_EPI_BIAS_F32 or dtype == F32
    for t_ in (a, w, resid):
        if t_ is not None and t_.dtype != storage_dtype(dtype):
            raise TypeError(f"gemm: operand dtype {t_.dtype} does not match storage dtype {storage_dtype(dtype)}")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32 if out_f32 else storage_dtype(dtype), device=a.device)
    ldc = n if ldc is None else ldc
    ldr = (n if ldr is None else ldr) if resid is not None else 0
    with L.on(a.device):
        L.check(L.lib().ag_gemm(L.ptr(a), lda, L.ptr(w), L.ptr(bias), L.ptr(out), ldc, L.ptr(resid), ldr,
                                rows_per_seq, resid_share, m, n, k, epilogue, dtype, L.ptr(ln_stats), L.ptr(ln_colsum),
                                float(ln_eps), L.ptr(stats_out), L.ptr(rows_dev), L.stream()))
    return out


NT, NN, TN = (0, 0), (0, 1), (1, 1)   # ag_gemm_ex operand orders: Linear forward / dX / dW


def gemm_ex_splits(m: int, n: int, kc: int) -> int:
    return int(L.lib().ag_gemm_ex_splits(m, n, kc))


def gemm_ex(a: Tensor, b: Tensor, order: Tuple[int, int] = NT, epilogue: int = L.AG_EX_STORE, bias: Optional[Tensor] = None,
            out_dtype: int = BF16, aux: Optional[Tensor] = None, splits: Optional[int] = None, out: Optional[Tensor] = None):
    """ag_gemm_ex on bf16 operands read in place.  order NT: a [M,Kc], b [N,Kc]; NN: a [M,Kc], b [Kc,N]; TN: a [Kc,M], b [Kc,N].
    -> AG_EX_STORE: C [M,N] (out_dtype); AG_EX_GELU_DUAL: (pre bf16, gelu(pre) bf16); AG_EX_GELU_BWD: bf16 acc * gelu'(aux);
    AG_EX_SLABS: fp32 [splits, M, N] partial sums (splits None: the library's recommendation)."""
    L.require_gpu(a, b, bias, aux, out)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise TypeError("gemm_ex: operands must be bf16")
    a_col, b_col = order
    (kc, m) = a.shape if a_col else a.shape[::-1]
    (kc2, n) = b.shape if b_col else b.shape[::-1]
    if kc != kc2:
        raise ValueError(f"gemm_ex: contraction lengths differ ({kc} vs {kc2})")
    lda, ldb = a.stride(0), b.stride(0)
    dev = a.device
    out2 = None
    slabs = None
    if epilogue == L.AG_EX_SLABS:
        splits = gemm_ex_splits(m, n, kc) if splits is None else splits
        slabs = out if out is not None else torch.empty((splits, m, n), dtype=torch.float32, device=dev)
        c, ldc = None, n
    else:
        splits = 1
        c = out if out is not None else torch.empty((m, n), dtype=storage_dtype(out_dtype), device=dev)
        ldc = c.stride(0)
        if epilogue == L.AG_EX_GELU_DUAL:
            out2 = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with L.on(dev):
        L.check(L.lib().ag_gemm_ex(L.ptr(a), lda, a_col, L.ptr(b), ldb, b_col, m, n, kc, epilogue, L.ptr(bias), L.ptr(c), ldc,
                                   out_dtype, L.ptr(aux), aux.stride(0) if aux is not None else 0, L.ptr(out2), n, splits,
                                   L.ptr(slabs), L.stream()))
    if epilogue == L.AG_EX_SLABS:
        return slabs
    if epilogue == L.AG_EX_GELU_DUAL:
        return c, out2
    return c


_SCRATCH = {}


def _scratch(dev, floats: int) -> Tensor:
    """a grow-only fp32 scratch buffer per device AND stream (per-block partials of the row kernels, consumed inside the call that
    fills it: stream order protects it on one stream — the backward's side streams run such kernels concurrently with the main
    stream's, so each stream has its own)."""
    key = (str(dev), L.stream())
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=dev)
        _SCRATCH[key] = buf
    return buf


def rows_finish(x: Tensor, bias: Optional[Tensor] = None, p_drop: float = 0.0, seed: int = 0, resid: Optional[Tensor] = None,
                ln: Optional[Tuple[Tensor, Tensor, float]] = None, want_t: bool = False, want_f32: bool = False,
                want_bf16: bool = True):
    """ag_rows_finish.  x: fp32 [M,H] or split-K slabs [S,M,H].  t = resid + dropout(x + bias); z = LayerNorm(t) (ln None: z = t).
    -> (t fp32 or None, z fp32 or None, z bf16 or None)."""
    L.require_gpu(x, bias, resid)
    splits = x.shape[0] if x.dim() == 3 else 1
    m, h = x.shape[-2], x.shape[-1]
    dev = x.device
    t = torch.empty((m, h), dtype=torch.float32, device=dev) if want_t else None
    zf = torch.empty((m, h), dtype=torch.float32, device=dev) if want_f32 else None
    zb = torch.empty((m, h), dtype=torch.bfloat16, device=dev) if want_bf16 else None
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    with L.on(dev):
        L.check(L.lib().ag_rows_finish(L.ptr(x), splits, m * h, L.ptr(bias), float(p_drop), seed & 0xFFFFFFFF, L.ptr(resid), L.ptr(t),
                                       L.ptr(g), L.ptr(b), float(eps), L.ptr(zf), L.ptr(zb), m, h, L.stream()))
    return t, zf, zb


def rows_ln_bwd(dy: Tensor, x: Optional[Tensor] = None, gamma: Optional[Tensor] = None, eps: float = 0.0,
                dy_add: Optional[Tensor] = None, add: Optional[Tensor] = None, want_dx: bool = True, want_bf16: bool = False,
                p_drop: float = 0.0, seed: int = 0, dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None,
                dbias: Optional[Tensor] = None, accumulate: bool = False):
    """ag_rows_ln_bwd.  dy: fp32 [M,H] or slabs [S,M,H] (+ dy_add); dx = LayerNorm backward at rows x (x None: dx = dy) (+ add).
    -> (dx fp32 or None, bf16(dropout'(dx)) or None); dgamma / dbeta / dbias [H] are written (or accumulated) when given."""
    L.require_gpu(dy, x, gamma, dy_add, add)
    splits = dy.shape[0] if dy.dim() == 3 else 1
    m, h = dy.shape[-2], dy.shape[-1]
    dev = dy.device
    dx = torch.empty((m, h), dtype=torch.float32, device=dev) if want_dx else None
    dxb = torch.empty((m, h), dtype=torch.bfloat16, device=dev) if want_bf16 else None
    need = dgamma is not None or dbeta is not None or dbias is not None
    scratch = _scratch(dev, int(L.lib().ag_rows_ln_bwd_scratch_floats(m, h))) if need else None
    with L.on(dev):
        L.check(L.lib().ag_rows_ln_bwd(L.ptr(dy), splits, m * h, L.ptr(dy_add), L.ptr(x), L.ptr(gamma), float(eps), L.ptr(add), L.ptr(dx),
                                       L.ptr(dxb), float(p_drop), seed & 0xFFFFFFFF, L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias),
                                       1 if accumulate else 0, L.ptr(scratch), m, h, L.stream()))
    return dx, dxb


def slab_reduce(slabs: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """sum of split-K slabs [S, ...] -> fp32 [...] (written into ``out`` / added to it)."""
    L.require_gpu(slabs, out)
    splits = slabs.shape[0]
    n = slabs[0].numel()
    if out is None:
        out = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
    with L.on(slabs.device):
        L.check(L.lib().ag_slab_reduce(L.ptr(slabs), splits, n, n, L.ptr(out), 1 if accumulate else 0, L.stream()))
    return out


def colsum_bf16(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """column sums of a bf16 [M,N] matrix -> fp32 [N]."""
    L.require_gpu(x, out)
    m, n = x.shape
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
    with L.on(x.device):
        L.check(L.lib().ag_colsum_bf16(L.ptr(x), m, n, x.stride(0), L.ptr(out), 1 if accumulate else 0, None, L.stream()))
    return out


def gemm_ex_group(pairs, order: Tuple[int, int] = TN, out_dtype: int = F32):
    """[(a, b)] bf16 operand pairs of ONE operand order -> [C_i] (fresh fp32 / bf16 tensors) by ag_gemm_ex_group: one launch per 8
    products, plain store, no bias.  order TN: a [Kc, M], b [Kc, N] -> C [M, N] (the dW products dY^T X of a backward)."""
    import ctypes as C
    if not pairs:
        return []
    a_col, b_col = order
    n = len(pairs)
    L.require_gpu(*[t_ for pr in pairs for t_ in pr])
    ms, ns, ks = [], [], []
    for a, b in pairs:
        if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
            raise TypeError("gemm_ex_group: operands must be bf16")
        (kc, m) = a.shape if a_col else a.shape[::-1]
        (kc2, nn_) = b.shape if b_col else b.shape[::-1]
        if kc != kc2:
            raise ValueError(f"gemm_ex_group: contraction lengths differ ({kc} vs {kc2})")
        ms.append(m); ns.append(nn_); ks.append(kc)
    dev = pairs[0][0].device
    outs = [torch.empty((m, nn_), dtype=storage_dtype(out_dtype), device=dev) for m, nn_ in zip(ms, ns)]
    pa = (C.c_void_p * n)(*[a.data_ptr() for a, _ in pairs])
    pb = (C.c_void_p * n)(*[b.data_ptr() for _, b in pairs])
    pc = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    lda = (C.c_int64 * n)(*[a.stride(0) for a, _ in pairs])
    ldb = (C.c_int64 * n)(*[b.stride(0) for _, b in pairs])
    ldc = (C.c_int64 * n)(*[o.stride(0) for o in outs])
    am, an, ak = (C.c_int * n)(*ms), (C.c_int * n)(*ns), (C.c_int * n)(*ks)
    with L.on(dev):
        L.check(L.lib().ag_gemm_ex_group(n, pa, lda, pb, ldb, am, an, ak, pc, ldc, a_col, b_col, out_dtype, L.stream()))
    return outs


def colsum_bf16_group(xs) -> list:
    """column sums of several bf16 [M_i, N_i] matrices -> [fp32 [N_i]] in one launch per 16 (ag_colsum_bf16_group; the bits of colsum_bf16)."""
    import ctypes as C
    if not xs:
        return []
    L.require_gpu(*xs)
    n = len(xs)
    dev = xs[0].device
    outs = [torch.empty(x.shape[1], dtype=torch.float32, device=dev) for x in xs]
    px = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    po = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    am, an = (C.c_int * n)(*[x.shape[0] for x in xs]), (C.c_int * n)(*[x.shape[1] for x in xs])
    ld = (C.c_int64 * n)(*[x.stride(0) for x in xs])
    with L.on(dev):
        L.check(L.lib().ag_colsum_bf16_group(n, px, am, an, ld, po, L.stream()))
    return outs


def cast_many(pairs) -> None:
    """[(src fp32 tensor, dst bf16 or fp32 tensor of the same numel)] -> one ag_cast_f32_many launch (per 96 segments)."""
    import ctypes as C
    pairs = [(s_, d_) for s_, d_ in pairs if s_.numel() > 0]
    if not pairs:
        return
    n = len(pairs)
    L.require_gpu(*[t_ for pr in pairs for t_ in pr])
    src = (C.c_void_p * n)(*[s_.data_ptr() for s_, _ in pairs])
    dst = (C.c_void_p * n)(*[d_.data_ptr() for _, d_ in pairs])
    cnt = (C.c_int64 * n)(*[s_.numel() for s_, _ in pairs])
    dty = (C.c_int * n)(*[(F32 if d_.dtype == torch.float32 else BF16) for _, d_ in pairs])
    for s_, d_ in pairs:
        if s_.dtype != torch.float32 or not s_.is_contiguous() or not d_.is_contiguous() or d_.numel() != s_.numel():
            raise ValueError("cast_many: sources must be contiguous fp32, destinations contiguous with the same element count")
    with L.on(pairs[0][0].device):
        L.check(L.lib().ag_cast_f32_many(src, dst, cnt, dty, n, L.stream()))


def pack_many(pairs, scale: float = 1.0) -> None:
    """cast_many with every element scaled on the way (ag_pack_f32_many): the gradients of one exchange bucket -> its flat send buffer."""
    import ctypes as C
    pairs = [(s_, d_) for s_, d_ in pairs if s_.numel() > 0]
    if not pairs:
        return
    n = len(pairs)
    L.require_gpu(*[t_ for pr in pairs for t_ in pr])
    for s_, d_ in pairs:
        if s_.dtype != torch.float32 or not s_.is_contiguous() or not d_.is_contiguous() or d_.numel() != s_.numel():
            raise ValueError("pack_many: sources must be contiguous fp32, destinations contiguous with the same element count")
    src = (C.c_void_p * n)(*[s_.data_ptr() for s_, _ in pairs])
    dst = (C.c_void_p * n)(*[d_.data_ptr() for _, d_ in pairs])
    cnt = (C.c_int64 * n)(*[s_.numel() for s_, _ in pairs])
    dty = (C.c_int * n)(*[(F32 if d_.dtype == torch.float32 else BF16) for _, d_ in pairs])
    with L.on(pairs[0][0].device):
        L.check(L.lib().ag_pack_f32_many(src, dst, cnt, dty, n, float(scale), L.stream()))


def set_dropout_salt(salt: int, device) -> None:
    """ag_set_dropout_salt on the current stream of ``device`` (0 = the eager default: seeds as given)."""
    with L.on(device):
        L.check(L.lib().ag_set_dropout_salt(salt & 0xFFFFFFFF, L.stream()))


def pad_cols(src: Tensor, cols_dst: int, dtype: int = F32) -> Tensor:
    """fp32 [M, ld >= C] (first C columns) -> dense [M, cols_dst] fp32 / bf16, zero-filled beyond C (or cut to cols_dst < C)."""
    L.require_gpu(src)
    m = src.shape[0]
    out = torch.empty((m, cols_dst), dtype=storage_dtype(dtype), device=src.device)
    with L.on(src.device):
        L.check(L.lib().ag_pad_cols_f32(L.ptr(src), src.stride(0), src.shape[1], L.ptr(out), cols_dst, cols_dst, dtype, m, L.stream()))
    return out


def masked_attention_train_bf16(qkv: Tensor, mask_bits: Tensor, rows: int, t: int, h: int, heads: int, mask_mode: int,
                                p_drop: float = 0.0, seed: int = 0) -> Tensor:
    """bf16 qkv [rows*t, 3h] -> bf16 ctx [rows*t, h] (MFMA attention of the bf16 training step; head dim 64, t <= 256)."""
    L.require_gpu(qkv, mask_bits)
    ctx = torch.empty((rows * t, h), dtype=torch.bfloat16, device=qkv.device)
    with L.on(qkv.device):
        L.check(L.lib().ag_masked_attention_train_bf16(L.ptr(qkv), L.ptr(mask_bits), L.ptr(ctx), rows, t, h, heads, mask_mode,
                                                       float(p_drop), seed & 0xFFFFFFFF, L.stream()))
    return ctx


def masked_attention_bwd_bf16(qkv: Tensor, mask_bits: Tensor, dctx: Tensor, rows: int, t: int, h: int, heads: int, mask_mode: int,
                              p_drop: float = 0.0, seed: int = 0) -> Tensor:
    """dqkv bf16 [rows*t, 3h] from dctx fp32 [rows*t, h] or split-K slabs [S, rows*t, h]."""
    L.require_gpu(qkv, mask_bits, dctx)
    splits = dctx.shape[0] if dctx.dim() == 3 else 1
    dqkv = torch.empty((rows * t, 3 * h), dtype=torch.bfloat16, device=qkv.device)
    with L.on(qkv.device):
        L.check(L.lib().ag_masked_attention_bwd_bf16(L.ptr(qkv), L.ptr(mask_bits), L.ptr(dctx), splits, rows * t * h, L.ptr(dqkv), rows, t, h,
                                                     heads, mask_mode, float(p_drop), seed & 0xFFFFFFFF, L.stream()))
    return dqkv


def gemm_resid_ln(a: Tensor, w: Tensor, bias: Optional[Tensor], r_pre: Tensor, r_stats: Tensor, ln_g: Tensor, ln_b: Tensor,
                  ln_eps: float, stats_out: Optional[Tensor] = None, rows_dev: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """A @ W^T + bias + LayerNorm(r_pre) with the LayerNorm recomputed in the epilogue from r_pre's slab statistics
    (ag_gemm_resid_ln; bf16, ring-kernel shapes only) -> (out [M,N] bf16, its slab statistics)."""
    L.require_gpu(a, w, bias, r_pre, r_stats, ln_g, ln_b)
    n, k = w.shape
    m = a.numel() // k
    out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    if stats_out is None:
        stats_out = new_row_stats(m, n, a.device)
    with L.on(a.device):
        L.check(L.lib().ag_gemm_resid_ln(L.ptr(a), k, L.ptr(w), L.ptr(bias), L.ptr(out), n, L.ptr(r_pre), n, L.ptr(r_stats), L.ptr(ln_g),
                                         L.ptr(ln_b), float(ln_eps), m, n, k, L.ptr(stats_out), L.ptr(rows_dev), L.stream()))
    return out, stats_out


def gemm_resid_ln_ws(a: Tensor, w: Tensor, bias: Optional[Tensor], r_pre: Tensor, r_stats: Tensor, ln_g: Tensor, ln_b: Tensor,
                     ln_eps: float, stats_out: Optional[Tensor] = None, rows_dev: Optional[Tensor] = None, m_expected: int = 0,
                     route: int = -1, splits: int = 0) -> Tuple[Tensor, Tensor]:
    """gemm_resid_ln, planned (ag_gemm_resid_ln_ws): route -1 = the library's cost model, WS_GEMM = the persistent kernel, WS_EX_SLABS =
    128 x 128 units x `splits` contraction ranges + the row kernel.  -> (out [M,N] bf16, its 256-column slab statistics)."""
    L.require_gpu(a, w, bias, r_pre, r_stats, ln_g, ln_b)
    n, k = w.shape
    m = a.numel() // k
    out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    if stats_out is None:
        stats_out = new_row_stats(m, n, a.device)
    with L.on(a.device):
        need = int(L.lib().ag_gemm_ws_scratch_bytes(m, n, k, L.AG_EPI_BIAS_RESID))
        if route >= 0:
            need = max(need, max(1, splits if splits > 0 else 8) * m * n * 4)
        scratch = _scratch(a.device, (need + 3) // 4) if need else None
        L.check(L.lib().ag_gemm_resid_ln_ws(L.ptr(a), k, L.ptr(w), L.ptr(bias), L.ptr(out), n, L.ptr(r_pre), n, L.ptr(r_stats), L.ptr(ln_g),
                                            L.ptr(ln_b), float(ln_eps), m, n, k, L.ptr(stats_out), L.ptr(rows_dev), int(m_expected), route, splits,
                                            L.ptr(scratch), scratch.numel() * 4 if scratch is not None else 0, L.stream()))
    return out, stats_out


def gemm_resid_split_scratch_bytes(m: int, n: int, k: int) -> int:
    """bytes of scratch ag_gemm_resid_split needs for this shape on the current device; 0: the shape does not split (use gemm)."""
    return int(L.lib().ag_gemm_resid_split_scratch_bytes(m, n, k))


def gemm_resid_split(a: Tensor, w: Tensor, bias: Optional[Tensor], resid: Tensor, stats_out: Optional[Tensor] = None,
                     out: Optional[Tensor] = None) -> Tensor:
    """A @ W^T + bias + resid (bf16, identity residual rows) with the under-filled tail round of tiles split over the contraction
    (ag_gemm_resid_split); raises when the shape does not split."""
    L.require_gpu(a, w, bias, resid, stats_out, out)
    n, k = w.shape
    m = a.numel() // k
    with L.on(a.device):
        need = gemm_resid_split_scratch_bytes(m, n, k)
        scratch = _scratch(a.device, (need + 3) // 4)
        if out is None:
            out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
        L.check(L.lib().ag_gemm_resid_split(L.ptr(a), a.stride(0), L.ptr(w), L.ptr(bias), L.ptr(out), out.stride(0), L.ptr(resid), resid.stride(0),
                                            m, n, k, L.ptr(stats_out), L.ptr(scratch), scratch.numel() * 4, L.stream()))
    return out


WS_GEMM, WS_BIG_SPLIT, WS_EX, WS_EX_SLABS = 0, 1, 2, 3   # routes of ag_gemm_ws (csrc/common.h)


def gemm_ws(a: Tensor, w: Tensor, bias: Optional[Tensor], epilogue: int, m: Optional[int] = None, lda: Optional[int] = None,
            resid: Optional[Tensor] = None, ldr: Optional[int] = None, rows_per_seq: int = 1, resid_share: int = 1,
            out: Optional[Tensor] = None, ldc: Optional[int] = None, ln_stats: Optional[Tensor] = None, stats_in_cols: int = 256,
            ln_colsum: Optional[Tensor] = None, ln_eps: float = 0.0, stats_out: Optional[Tensor] = None, out_cols_ok: int = 3,
            rows_dev: Optional[Tensor] = None, route: int = -1, splits: int = 0):
    """ag_gemm_ws: the planned Linear of the masked forward at under-filled launch sizes (bf16).  -> (out, slab width of the
    statistics written or 0).  route < 0: planned by the library's cost model; else WS_GEMM / WS_BIG_SPLIT / WS_EX / WS_EX_SLABS."""
    L.require_gpu(a, w, bias, resid, out, ln_stats, ln_colsum, stats_out)
    n, k = w.shape
    m = a.numel() // k if m is None else m
    lda = k if lda is None else lda
    if out is None:
        out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    ldc = n if ldc is None else ldc
    ldr = (n if ldr is None else ldr) if resid is not None else 0
    cols = C.c_int(0)
    with L.on(a.device):
        need = int(L.lib().ag_gemm_ws_scratch_bytes(m, n, k, epilogue))
        if route >= 0:        # (a pinned route / split count — the parity tests — may ask for more slabs than the planner would)
            need = max(need, max(1, splits if splits > 0 else 8) * m * n * 4)
        scratch = _scratch(a.device, (need + 3) // 4) if need else None
        L.check(L.lib().ag_gemm_ws(L.ptr(a), lda, L.ptr(w), L.ptr(bias), L.ptr(out), ldc, L.ptr(resid), ldr, rows_per_seq, resid_share,
                                   m, n, k, epilogue, L.AG_BF16, L.ptr(ln_stats), stats_in_cols, L.ptr(ln_colsum), float(ln_eps),
                                   L.ptr(stats_out), out_cols_ok, C.byref(cols), L.ptr(rows_dev), route, splits,
                                   L.ptr(scratch), scratch.numel() * 4 if scratch is not None else 0, L.stream()))
    return out, int(cols.value)


def side_mlp(x: Tensor, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], ln_g: Optional[Tensor],
             ln_b: Optional[Tensor], eps: float, post_ln: bool, rows_dev: Optional[Tensor] = None) -> Tensor:
    """fused MLP half of a narrow layer (ag_side_mlp): x [M, h] bf16 -> [M, h] bf16."""
    L.require_gpu(x, w1, w2, b1, b2, ln_g, ln_b)
    x = x.contiguous()
    m, h = x.shape
    out = torch.empty_like(x)
    with L.on(x.device):
        L.check(L.lib().ag_side_mlp(L.ptr(x), h, m, h, w1.shape[0], L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(ln_g), L.ptr(ln_b),
                                    float(eps), 1 if post_ln else 0, L.ptr(out), h, L.ptr(rows_dev), L.stream()))
    return out


def side_linear(x: Tensor, w: Tensor, b: Optional[Tensor], pre_ln=None, resid: Optional[Tensor] = None, post_ln=None,
                eps: float = 1e-12, rows_dev: Optional[Tensor] = None) -> Tensor:
    """fused narrow Linear (ag_side_linear): LN_post(resid + W . LN_pre(x) + b); pre_ln / post_ln = (gamma, beta) or None."""
    L.require_gpu(x, w, b, resid)
    x = x.contiguous()
    m, h = x.shape
    n = w.shape[0]
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    g0, b0 = pre_ln if pre_ln is not None else (None, None)
    g1, b1 = post_ln if post_ln is not None else (None, None)
    r = resid.contiguous() if resid is not None else None
    with L.on(x.device):
        L.check(L.lib().ag_side_linear(L.ptr(x), h, m, h, n, L.ptr(w), L.ptr(b), L.ptr(g0), L.ptr(b0), L.ptr(r), n, L.ptr(g1), L.ptr(b1),
                                       float(eps), L.ptr(out), n, L.ptr(rows_dev), L.stream()))
    return out


def stat_slabs(h: int) -> int:
    return (h + 255) // 256


def new_row_stats(rows: int, h: int, device) -> Tensor:
    """uninitialised row-statistics buffer [ceil(h/256), rows, 2] (include/autognothi_hip.h: AG_ROW_STATS_FLOATS)."""
    return torch.empty((stat_slabs(h), rows, 2), dtype=torch.float32, device=device)


def reduce_row_stats(st: Tensor, rows: int, h: int) -> Tensor:
    """[S, rows, 2] slab partials -> [rows, 2] totals (a view helper for tests / diagnostics; the kernels add the slabs
    themselves)."""
    return st.view(stat_slabs(h), rows, 2).sum(dim=0)


def row_stats(x: Tensor) -> Tensor:
    """per-256-column partial (sum, sum of squares) of each row of a bf16 [rows, H] tensor -> fp32 [ceil(H/256), rows, 2]."""
    L.require_gpu(x)
    x = x.contiguous()
    rows, h = x.shape
    st = new_row_stats(rows, h, x.device)
    with L.on(x.device):
        L.check(L.lib().ag_row_stats_bf16(L.ptr(x), h, rows, h, L.ptr(st), L.stream()))
    return st


def masked_attention(qkv: Tensor, mask_bits: Tensor, rows: int, t: int, h: int, heads: int, share: int,
                     mask_mode: int, dtype: int, n_query: int = 0) -> Tensor:
    L.require_gpu(qkv, mask_bits)
    ctx = torch.empty((rows, t, h), dtype=storage_dtype(dtype), device=qkv.device)
    if n_query:
        ctx.zero_()
    with L.on(qkv.device):
        L.check(L.lib().ag_masked_attention(L.ptr(qkv), L.ptr(mask_bits), L.ptr(ctx), rows, t, h, heads, share,
                                            mask_mode, n_query, dtype, L.stream()))
    return ctx


def softmax_rows(x: Tensor) -> Tensor:
    L.require_gpu(x)
    x = x.contiguous().float()
    y = torch.empty_like(x)
    with L.on(x.device):
        L.check(L.lib().ag_softmax_rows(L.ptr(x), L.ptr(y), x.shape[0], x.shape[1], L.stream()))
    return y


# ----------------------------------------------------------------------------- Shapley reductions
def shapley_normalize(pred: Tensor, grand: Optional[Tensor], null: Optional[Tensor], normalize: bool = True) -> Tensor:
    """pred [B,T,C] (T includes CLS) -> phi [B,C,P] (models/shapley.py:82-93 + vanilla_vit.py:129)."""
    L.require_gpu(pred, grand, null)
    pred = pred.contiguous().float()
    b, t, c = pred.shape
    phi = torch.empty((b, c, t - 1), dtype=torch.float32, device=pred.device)
    g = grand.contiguous().float() if grand is not None else None
    n = null.contiguous().float() if null is not None else None
    with L.on(pred.device):
        L.check(L.lib().ag_shapley_normalize(L.ptr(pred), L.ptr(g), L.ptr(n), b, t, c, 1 if normalize else 0,
                                             L.ptr(phi), L.stream()))
    return phi


def shapley_normalize_bwd(dphi: Tensor, t: int, normalize: bool = True) -> Tensor:
    L.require_gpu(dphi)
    dphi = dphi.contiguous().float()
    b, c, p = dphi.shape
    dpred = torch.empty((b, t, c), dtype=torch.float32, device=dphi.device)
    with L.on(dphi.device):
        L.check(L.lib().ag_shapley_normalize_bwd(L.ptr(dphi), b, t, c, 1 if normalize else 0, L.ptr(dpred), L.stream()))
    return dpred


def shapley_loss(mask_bits: Tensor, v0: Tensor, vs: Tensor, phi: Tensor, batch: int, k: int, want_grad: bool = True):
    """models/shapley.py:9-53 -> (loss [1] fp32 on device, dphi [B,C,P] or None)."""
    L.require_gpu(mask_bits, v0, vs, phi)
    phi = phi.contiguous().float()
    b, c, p = phi.shape
    assert b == batch and vs.shape[0] == batch * k
    loss = torch.empty(1, dtype=torch.float32, device=phi.device)
    dphi = torch.empty_like(phi) if want_grad else None
    scratch = torch.empty(batch * k * c, dtype=torch.float32, device=phi.device)
    with L.on(phi.device):
        L.check(L.lib().ag_shapley_loss(L.ptr(mask_bits), L.ptr(v0.contiguous().float()), L.ptr(vs.contiguous().float()),
                                        L.ptr(phi), batch, k, p, c, L.ptr(loss), L.ptr(dphi), L.ptr(scratch), L.stream()))
    return loss, dphi


def kl_loss(ref: Tensor, cur: Tensor, want_grad: bool = True):
    """models/shapley.py:96-106 -> (loss [1], dloss/dcur or None)."""
    L.require_gpu(ref, cur)
    ref, cur = ref.contiguous().float(), cur.contiguous().float()
    loss = torch.empty(1, dtype=torch.float32, device=ref.device)
    d = torch.empty_like(cur) if want_grad else None
    with L.on(ref.device):
        L.check(L.lib().ag_kl_loss(L.ptr(ref), L.ptr(cur), ref.shape[0], ref.shape[1], L.ptr(loss), L.ptr(d), L.stream()))
    return loss, d


def seq_compact_plan(mask_bits: Tensor, t: int, sync: bool = True):
    """BERT token pruning plan: key bits [R, Tw] -> (cu_seqlens int32 [R+1], packed-row source table int32 [R*t], N).
    sync=True reads the packed row count N back (a 4-byte device->host copy); sync=False returns the upper bound R*t
    instead: pass ``rows_dev=cu[R:R+1]`` to the ops of the packed section and nothing leaves the device."""
    L.require_gpu(mask_bits)
    rows = mask_bits.shape[0]
    cu = torch.empty(rows + 1, dtype=torch.int32, device=mask_bits.device)
    src = torch.empty(rows * t, dtype=torch.int32, device=mask_bits.device)
    with L.on(mask_bits.device):
        L.check(L.lib().ag_seq_compact_plan(L.ptr(mask_bits.contiguous()), rows, t, L.ptr(cu), L.ptr(src), L.stream()))
    return cu, src, (int(cu[rows].item()) if sync else rows * t)


_HIP_RT = None


_CU_PARTITIONS = {}


def cu_partition_streams(device, cus_per_xcd_first: int):
    """Two HIP streams that share the device's CUs without overlapping: the first runs on CUs [0, c) of EVERY XCD, the second on the
    rest (hipExtStreamCreateWithCUMask; on MI355X mask bit i is CU i // 8 of XCD i % 8, measured with tools/probe/cumask_probe.cpp).
    The library is told each stream's CU count (ag_set_stream_cus: the persistent GEMM sizes its grid by it).
    -> (torch stream A, torch stream B, CUs of A, CUs of B)."""
    global _HIP_RT
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(cus_per_xcd_first))
    if key in _CU_PARTITIONS:        # one pair per (device, split), for the life of the process: a CU-masked stream is never destroyed here, so
        return _CU_PARTITIONS[key]   # its handle — the key of the library's CU-budget table — is never reused for another stream
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    per_xcd = n_cu // 8
    c = int(cus_per_xcd_first)
    if not (0 < c < per_xcd) or n_cu % 8 != 0:
        raise ValueError(f"cu_partition_streams: need 0 < {c} < {per_xcd} CUs per XCD on a device whose CU count is a multiple of 8")
    if _HIP_RT is None:
        _HIP_RT = C.CDLL("libamdhip64.so")
    words = (n_cu + 31) // 32
    out = []
    with torch.cuda.device(dev):
        for lo, hi in ((0, 8 * c), (8 * c, n_cu)):
            mask = (C.c_uint32 * words)()
            for i in range(lo, hi):
                mask[i // 32] |= 1 << (i % 32)
            handle = C.c_void_p()
            rc = _HIP_RT.hipExtStreamCreateWithCUMask(C.byref(handle), C.c_uint32(words), mask)
            if rc != 0 or not handle.value:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
            L.check(L.lib().ag_set_stream_cus(handle, hi - lo))
            out.append(torch.cuda.ExternalStream(handle.value, device=dev))
    _CU_PARTITIONS[key] = (out[0], out[1], 8 * c, n_cu - 8 * c)
    return _CU_PARTITIONS[key]


def reload_knobs() -> None:
    """have the library read its experiment / test knobs (AG_GEMM_* ...) from the environment again (they are cached)."""
    L.check(L.lib().ag_reload_knobs())


def gather_rows(src: Tensor, index: Tensor, n: int, dtype: int, rows_dev: Optional[Tensor] = None) -> Tensor:
    """dst[i, :] = src2d[index[i], :] for the first n entries of index (src viewed as [-1, H])."""
    L.require_gpu(src, index)
    h = src.shape[-1]
    s2 = src.contiguous().view(-1, h)
    out = torch.empty((n, h), dtype=s2.dtype, device=s2.device)
    with L.on(s2.device):
        L.check(L.lib().ag_gather_rows(L.ptr(s2), h, L.ptr(index), L.ptr(out), h, n, h, dtype, L.ptr(rows_dev), L.stream()))
    return out


def mc_shapley_reduce(v: Tensor, rank: Tensor):
    """scripts/preview_text_shapley.py:112-153: v [reps, P+1, C] fp32, rank [reps, P] int32 -> (sv [C,P], v0 [C], vn [C])."""
    L.require_gpu(v, rank)
    v, rank = v.contiguous().float(), rank.contiguous().to(torch.int32)
    reps, p1, c = v.shape
    sv = torch.empty((c, p1 - 1), dtype=torch.float32, device=v.device)
    v0 = torch.empty(c, dtype=torch.float32, device=v.device)
    vn = torch.empty(c, dtype=torch.float32, device=v.device)
    with L.on(v.device):
        L.check(L.lib().ag_mc_shapley_reduce(L.ptr(v), L.ptr(rank), reps, p1 - 1, c, L.ptr(sv), L.ptr(v0), L.ptr(vn), L.stream()))
    return sv, v0, vn


# ----------------------------------------------------------------------------- training building blocks (fp32)
def _f32c(t: Tensor) -> Tensor:
    L.require_gpu(t)
    if t.dtype != torch.float32:
        raise TypeError(f"training kernels are fp32; got {t.dtype}")
    return t.contiguous()


def transpose(src: Tensor, pad_cols_to: int = 1) -> Tensor:
    """[R,C] -> [C, Rp] with Rp = R rounded up to `pad_cols_to` (zero padded) — the K-dim padding ag_gemm needs."""
    s = _f32c(src)
    r, c = s.shape
    rp = (r + pad_cols_to - 1) // pad_cols_to * pad_cols_to
    dst = torch.zeros((c, rp), dtype=torch.float32, device=s.device) if rp != r else torch.empty((c, rp), dtype=torch.float32, device=s.device)
    with L.on(s.device):
        L.check(L.lib().ag_transpose_f32(L.ptr(s), r, c, c, L.ptr(dst), rp, L.stream()))
    return dst


def transpose_bf16(src: Tensor, pad_cols_to: int = 1) -> Tensor:
    """[R,C] fp32 -> [C, Rp] bf16 (transpose + round in one pass, zero padded)."""
    s = _f32c(src)
    r, c = s.shape
    rp = (r + pad_cols_to - 1) // pad_cols_to * pad_cols_to
    dst = torch.empty((c, rp), dtype=torch.bfloat16, device=s.device)
    with L.on(s.device):
        L.check(L.lib().ag_transpose_f32_bf16(L.ptr(s), r, c, c, L.ptr(dst), rp, L.stream()))
    return dst


def cast_transpose_bf16(src: Tensor, pad_cols_to: int = 1) -> Tuple[Tensor, Tensor]:
    """[R,C] fp32 -> ([R,C] bf16, [C,Rp] bf16): both operand forms of a mixed-precision Linear in one launch."""
    s = _f32c(src)
    r, c = s.shape
    rp = (r + pad_cols_to - 1) // pad_cols_to * pad_cols_to
    plain = torch.empty((r, c), dtype=torch.bfloat16, device=s.device)
    dst = torch.empty((c, rp), dtype=torch.bfloat16, device=s.device)
    with L.on(s.device):
        L.check(L.lib().ag_cast_transpose_f32_bf16(L.ptr(s), r, c, c, L.ptr(plain), L.ptr(dst), rp, L.stream()))
    return plain, dst


def gelu_cast_transpose_bf16(u: Tensor, pad_cols_to: int = 1) -> Tuple[Tensor, Tensor]:
    """gelu(u) as ([R,C] bf16, [C,Rp] bf16) in one pass over u (the fp32 GELU output is never written)."""
    s = _f32c(u)
    r, c = s.shape
    rp = (r + pad_cols_to - 1) // pad_cols_to * pad_cols_to
    plain = torch.empty((r, c), dtype=torch.bfloat16, device=s.device)
    dst = torch.empty((c, rp), dtype=torch.bfloat16, device=s.device)
    with L.on(s.device):
        L.check(L.lib().ag_gelu_cast_transpose_f32_bf16(L.ptr(s), r, c, L.ptr(plain), L.ptr(dst), rp, L.stream()))
    return plain, dst


def gelu_bwd_cast_transpose_bf16(u: Tensor, dy: Tensor, pad_cols_to: int = 1, want_f32: bool = True):
    """du = dy * gelu'(u) -> (du fp32 or None, [R,C] bf16, [C,Rp] bf16) in one pass."""
    u, dy = _f32c(u), _f32c(dy)
    r, c = dy.shape
    rp = (r + pad_cols_to - 1) // pad_cols_to * pad_cols_to
    du = torch.empty_like(dy) if want_f32 else None
    plain = torch.empty((r, c), dtype=torch.bfloat16, device=dy.device)
    dst = torch.empty((c, rp), dtype=torch.bfloat16, device=dy.device)
    with L.on(dy.device):
        L.check(L.lib().ag_gelu_bwd_cast_transpose_f32_bf16(L.ptr(u), L.ptr(dy), r, c, L.ptr(du), L.ptr(plain), L.ptr(dst), rp, L.stream()))
    return du, plain, dst


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    x = _f32c(x)
    m, n = x.shape
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
        accumulate = False
    with L.on(x.device):
        L.check(L.lib().ag_colsum_f32(L.ptr(x), m, n, n, L.ptr(out), 1 if accumulate else 0, L.stream()))
    return out


def _unary(name: str, *tensors: Tensor) -> Tensor:
    ts = [_f32c(t) for t in tensors]
    out = torch.empty_like(ts[0])
    with L.on(out.device):
        L.check(getattr(L.lib(), name)(*[L.ptr(t) for t in ts], L.ptr(out), out.numel(), L.stream()))
    return out


def gelu(u: Tensor) -> Tensor:
    return _unary("ag_gelu_f32", u)


def gelu_bwd(u: Tensor, dy: Tensor) -> Tensor:
    return _unary("ag_gelu_bwd_f32", u, dy)


def tanh_bwd(y: Tensor, dy: Tensor) -> Tensor:
    return _unary("ag_tanh_bwd_f32", y, dy)


def add(a: Tensor, b: Tensor) -> Tensor:
    return _unary("ag_add_f32", a, b)


def dropout(x: Tensor, p: float, seed: int) -> Tensor:
    """y = keep(seed, i) ? x/(1-p) : 0.  The same call on dy is the backward."""
    if p <= 0.0:
        return x
    x = _f32c(x)
    y = torch.empty_like(x)
    with L.on(x.device):
        L.check(L.lib().ag_dropout_f32(L.ptr(x), L.ptr(y), x.numel(), float(p), seed & 0xFFFFFFFF, L.stream()))
    return y


def dropout_add(x: Tensor, resid: Tensor, p: float, seed: int) -> Tensor:
    """resid + dropout(x) in one pass (p = 0: the plain sum)."""
    x, resid = _f32c(x), _f32c(resid)
    y = torch.empty_like(x)
    with L.on(x.device):
        L.check(L.lib().ag_dropout_add_f32(L.ptr(x), L.ptr(resid), L.ptr(y), x.numel(), float(max(p, 0.0)), seed & 0xFFFFFFFF, L.stream()))
    return y


def softmax_rows_bwd(y: Tensor, dy: Tensor) -> Tensor:
    y, dy = _f32c(y), _f32c(dy)
    dx = torch.empty_like(y)
    with L.on(y.device):
        L.check(L.lib().ag_softmax_rows_bwd(L.ptr(y), L.ptr(dy), L.ptr(dx), y.shape[0], y.shape[1], L.stream()))
    return dx


def layernorm_bwd(x: Tensor, gamma: Optional[Tensor], dy: Tensor, eps: float, dgamma: Optional[Tensor] = None,
                  dbeta: Optional[Tensor] = None, accumulate: bool = True, add: Optional[Tensor] = None) -> Tensor:
    """x, dy [rows, H] fp32 -> dx (+ add: the gradient of the residual branch); dgamma / dbeta ([H]) accumulated in place when given."""
    x, dy = _f32c(x), _f32c(dy)
    add = _f32c(add) if add is not None else None
    h = x.shape[-1]
    rows = x.numel() // h
    dx = torch.empty_like(x)
    scratch = torch.empty(256 * 2 * h, dtype=torch.float32, device=x.device)
    with L.on(x.device):
        L.check(L.lib().ag_layernorm_bwd_add(L.ptr(x), L.ptr(gamma), L.ptr(dy), L.ptr(add), rows, h, eps, L.ptr(dx), L.ptr(dgamma),
                                             L.ptr(dbeta), 1 if accumulate else 0, L.ptr(scratch), L.stream()))
    return dx


def masked_attention_train(qkv: Tensor, mask_bits: Tensor, rows: int, t: int, h: int, heads: int, mask_mode: int,
                           p_drop: float = 0.0, seed: int = 0, mixed: bool = False) -> Tensor:
    qkv = _f32c(qkv)
    ctx = torch.empty((rows, t, h), dtype=torch.float32, device=qkv.device)
    if mixed and h == heads * 64 and t <= 256:
        with L.on(qkv.device):
            L.check(L.lib().ag_masked_attention_train_mixed(L.ptr(qkv), L.ptr(mask_bits), L.ptr(ctx), rows, t, h, heads, mask_mode,
                                                            float(p_drop), seed & 0xFFFFFFFF, L.stream()))
        return ctx
    with L.on(qkv.device):
        L.check(L.lib().ag_masked_attention_train(L.ptr(qkv), L.ptr(mask_bits), L.ptr(ctx), rows, t, h, heads, mask_mode,
                                                  float(p_drop), seed & 0xFFFFFFFF, L.stream()))
    return ctx


def masked_attention_bwd(qkv: Tensor, mask_bits: Tensor, ctx: Tensor, dctx: Tensor, rows: int, t: int, h: int, heads: int,
                         mask_mode: int, p_drop: float = 0.0, seed: int = 0, mixed: bool = False) -> Tensor:
    """dqkv [rows,t,3h] fp32.  ``mixed``: bf16 MFMA operands (head dim 64, t <= 256), else the exact-fp32 kernels."""
    qkv, ctx, dctx = _f32c(qkv), _f32c(ctx), _f32c(dctx)
    dqkv = torch.empty_like(qkv)
    if mixed and h == heads * 64 and t <= 256:
        with L.on(qkv.device):
            L.check(L.lib().ag_masked_attention_bwd_mixed(L.ptr(qkv), L.ptr(mask_bits), L.ptr(ctx), L.ptr(dctx), L.ptr(dqkv),
                                                          rows, t, h, heads, mask_mode, float(p_drop), seed & 0xFFFFFFFF, L.stream()))
        return dqkv
    stats = torch.empty(rows * heads * t * 3, dtype=torch.float32, device=qkv.device)
    with L.on(qkv.device):
        L.check(L.lib().ag_masked_attention_bwd(L.ptr(qkv), L.ptr(mask_bits), L.ptr(ctx), L.ptr(dctx), L.ptr(dqkv), L.ptr(stats),
                                                rows, t, h, heads, mask_mode, float(p_drop), seed & 0xFFFFFFFF, L.stream()))
    return dqkv
