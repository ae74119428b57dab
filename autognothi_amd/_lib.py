"""ctypes binding of the C ABI declared in ``include/autognothi_hip.h``.

The product path has no CPU or eager-PyTorch fallback: if ``lib/libautognothi_hip.so`` is
missing, or a compute op is called with non-GPU tensors, this raises.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` (or ``autognothi_amd/csrc/build.sh``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

def _hip_touched() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_initialized())
    except Exception:
        return False


# Did the host program use the GPU before this package was imported?  Then the package's background stream (below) cannot have been the
# process's first second stream, and whether it runs beside or behind the caller's is out of this package's hands — measured (round 5,
# tests/test_gpu_scripts.py): five foreign streams + one captured graph before the first launch: two-stream epoch 372 images/s against 510 on
# one stream, with two idle waves on the two streams running perfectly beside each other (the yes / no probe says yes: it sees the hardware
# queues, not what the dispatcher does with 400 dependent launches beside a persistent kernel).  The two-stream epoch is therefore the
# default only in a process where this import came first (scripts/common.train_partition); AG_TRAIN_PARTITION=<n> forces it.
HIP_TOUCHED_BEFORE_IMPORT = _hip_touched()

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AG_HIP_LIB") or os.path.join(_HERE, "lib", "libautognothi_hip.so")   # AG_HIP_LIB: A/B another build

AG_OK = 0
AG_F32, AG_BF16 = 0, 1
AG_MASK_VIT_MUL, AG_MASK_BERT_ADD = 0, 1
AG_EPI_BIAS, AG_EPI_BIAS_GELU, AG_EPI_BIAS_RESID, AG_EPI_BIAS_F32, AG_EPI_BIAS_TANH, AG_EPI_BIAS_GELU_ADD = 0, 1, 2, 3, 4, 5
AG_EX_STORE, AG_EX_GELU_DUAL, AG_EX_GELU_BWD, AG_EX_SLABS = 0, 1, 2, 3
AG_MT_STATE_BYTES = 2560

vp, i32, i64, u32, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_float, C.c_size_t


class ag_layer_weights(C.Structure):
    _fields_ = [(n, vp) for n in ("w_qkv", "b_qkv", "w_o", "b_o", "w_fc1", "b_fc1", "w_fc2", "b_fc2",
                                  "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                                  "w_qkv_ln", "b_qkv_ln", "s_qkv_ln", "w_fc1_ln", "b_fc1_ln", "s_fc1_ln")]


class ag_encoder_desc(C.Structure):
    _fields_ = [("kind", i32), ("dtype", i32), ("T", i32), ("H", i32), ("I", i32), ("heads", i32),
                ("ln_eps", f32), ("n_layers", i32), ("layers", C.POINTER(ag_layer_weights))]


# name -> (restype, argtypes); mirrors include/autognothi_hip.h one to one
SIGNATURES = {
    "ag_abi_version": (i32, []),
    "ag_last_error": (C.c_char_p, []),
    "ag_device_info": (i32, [i32, C.POINTER(i32), C.c_char_p, sz]),
    "ag_mt19937_seed": (i32, [vp, u32, vp]),
    "ag_mt19937_import": (i32, [vp, vp, i32, vp]),
    "ag_mt19937_export": (i32, [vp, vp, C.POINTER(i32), vp]),
    "ag_mt19937_skip": (i32, [vp, i64, vp]),
    "ag_mt19937_raw": (i32, [vp, vp, i64, vp]),
    "ag_mask_shapley_new": (i32, [vp, i32, i32, vp, vp, vp, vp, vp]),
    "ag_mask_shapley_new_rows": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "ag_mask_purely_uniform": (i32, [vp, i32, i32, vp, vp, vp, vp]),
    "ag_mask_purely_uniform_rows": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "ag_pack_mask": (i32, [vp, i32, i32, vp, vp]),
    "ag_perturbed_masks": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "ag_cast_f32": (i32, [vp, vp, i64, i32, vp]),
    "ag_pack_folded_linear": (i32, [vp, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp]),
    "ag_layernorm": (i32, [vp, i32, i64, i32, i32, vp, vp, f32, vp, vp, i32, vp, vp]),
    "ag_gemm": (i32, [vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp]),
    "ag_gemm_ex": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp, vp, i64, i32, vp, i64, vp, i64, i32, vp, vp]),
    "ag_gemm_ex_splits": (i32, [i32, i32, i32]),
    "ag_gemm_ex_group": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ag_rows_finish": (i32, [vp, i32, i64, vp, f32, u32, vp, vp, vp, vp, f32, vp, vp, i32, i32, vp]),
    "ag_rows_ln_bwd_scratch_floats": (sz, [i32, i32]),
    "ag_rows_ln_bwd": (i32, [vp, i32, i64, vp, vp, vp, f32, vp, vp, vp, f32, u32, vp, vp, vp, i32, vp, i32, i32, vp]),
    "ag_slab_reduce": (i32, [vp, i32, i64, i64, vp, i32, vp]),
    "ag_colsum_bf16_scratch_floats": (sz, [i32, i32]),
    "ag_colsum_bf16": (i32, [vp, i32, i32, i64, vp, i32, vp, vp]),
    "ag_colsum_bf16_group": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "ag_cast_f32_many": (i32, [vp, vp, vp, vp, i32, vp]),
    "ag_pack_f32_many": (i32, [vp, vp, vp, vp, i32, f32, vp]),
    "ag_set_dropout_salt": (i32, [u32, vp]),
    "ag_pad_cols_f32": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, vp]),
    "ag_masked_attention_train_bf16": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, u32, vp]),
    "ag_masked_attention_bwd_bf16": (i32, [vp, vp, vp, i32, i64, vp, i32, i32, i32, i32, i32, f32, u32, vp]),
    "ag_gemm_supports_ln_fold": (i32, [i32, i32, i32, i64, i64, i64, i32, i32]),
    "ag_gemm_resid_ln": (i32, [vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, vp, f32, i32, i32, i32, vp, vp, vp]),
    "ag_gemm_resid_ln_supported": (i32, [i32, i32, i32, i64, i64, i64]),
    "ag_gemm_resid_split_scratch_bytes": (C.c_size_t, [i32, i32, i32]),
    "ag_gemm_resid_split": (i32, [vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, vp, vp, C.c_size_t, vp]),
    "ag_gemm_resid_ln_ws": (i32, [vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, vp, f32, i32, i32, i32, vp, vp, i32, i32, i32, vp, C.c_size_t, vp]),
    "ag_gemm_ws_scratch_bytes": (C.c_size_t, [i32, i32, i32, i32]),
    "ag_gemm_ws": (i32, [vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, f32, vp, i32, vp, vp, i32, i32,
                         vp, C.c_size_t, vp]),
    "ag_side_mlp_supported": (i32, [i32, i32, i32]),
    "ag_side_mlp": (i32, [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, f32, i32, vp, i64, vp, vp]),
    "ag_side_linear_supported": (i32, [i32, i32, i32, i32]),
    "ag_side_linear": (i32, [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, f32, vp, i64, vp, vp]),
    "ag_row_stats_bf16": (i32, [vp, i64, i32, i32, vp, vp]),
    "ag_cls_last_is_supported": (i32, [i32, i32, i32, i32]),
    "ag_cls_last_workspace_bytes": (C.c_size_t, [i32, i32, i32]),
    "ag_cls_last_attention_rows": (i32, [vp, vp, i32, vp, vp, vp, vp, C.c_float, vp, i64, i32, i32, i32, i32, vp, C.c_size_t, vp]),
    "ag_masked_attention": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ag_vit_im2col": (i32, [vp, i32, i32, i32, i32, vp, i32, vp]),
    "ag_vit_assemble": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "ag_bert_embed": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, vp, vp, f32, vp, vp, i32, vp]),
    "ag_softmax_rows": (i32, [vp, vp, i32, i32, vp]),
    "ag_shapley_normalize": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "ag_shapley_normalize_bwd": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "ag_shapley_normalize_rows": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "ag_shapley_normalize_rows_bwd": (i32, [vp, i32, i32, i32, vp, vp]),
    "ag_shapley_loss": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "ag_kl_loss": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "ag_transpose_f32": (i32, [vp, i32, i32, i64, vp, i64, vp]),
    "ag_transpose_f32_bf16": (i32, [vp, i32, i32, i64, vp, i64, vp]),
    "ag_cast_transpose_f32_bf16": (i32, [vp, i32, i32, i64, vp, vp, i64, vp]),
    "ag_colsum_f32": (i32, [vp, i32, i32, i64, vp, i32, vp]),
    "ag_gelu_f32": (i32, [vp, vp, i64, vp]),
    "ag_gelu_bwd_f32": (i32, [vp, vp, vp, i64, vp]),
    "ag_tanh_bwd_f32": (i32, [vp, vp, vp, i64, vp]),
    "ag_add_f32": (i32, [vp, vp, vp, i64, vp]),
    "ag_dropout_add_f32": (i32, [vp, vp, vp, i64, f32, u32, vp]),
    "ag_gelu_cast_transpose_f32_bf16": (i32, [vp, i32, i32, vp, vp, i64, vp]),
    "ag_gelu_bwd_cast_transpose_f32_bf16": (i32, [vp, vp, i32, i32, vp, vp, vp, i64, vp]),
    "ag_layernorm_bwd_add": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, i32, vp, vp]),
    "ag_dropout_f32": (i32, [vp, vp, i64, f32, u32, vp]),
    "ag_softmax_rows_bwd": (i32, [vp, vp, vp, i32, i32, vp]),
    "ag_layernorm_bwd": (i32, [vp, vp, vp, i32, i32, f32, vp, vp, vp, i32, vp, vp]),
    "ag_masked_attention_train": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, u32, vp]),
    "ag_masked_attention_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, u32, vp]),
    "ag_masked_attention_train_mixed": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, u32, vp]),
    "ag_masked_attention_bwd_mixed": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, u32, vp]),
    "ag_mc_shapley_reduce": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "ag_probe_mfma": (i32, [i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]),
    "ag_probe_spin": (i32, [i32, vp]),
    "ag_probe_dma": (i32, [i32, i32, i32, vp, vp, i64, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), vp]),
    "ag_probe_store": (i32, [i32, i32, vp, i64, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]),
    "ag_launch_count": (i64, []),
    "ag_gemm_last_plan": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ag_profile_enable": (i32, [i32]),
    "ag_profile_collect": (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]),
    "ag_encoder_workspace_bytes": (sz, [C.POINTER(ag_encoder_desc), i32]),
    "ag_encoder_forward": (i32, [C.POINTER(ag_encoder_desc), vp, i32, i32, vp, vp, i32, vp, sz, vp]),
    "ag_encoder_forward_chained": (i32, [C.POINTER(ag_encoder_desc), vp, i32, i32, vp, vp, i32, vp, sz, vp, i32, i32, C.POINTER(i32), vp]),
    "ag_bert_encoder_forward_pruned": (i32, [C.POINTER(ag_encoder_desc), vp, i32, i32, vp, vp, vp, sz, vp, vp]),
    "ag_bert_layers_forward_packed": (i32, [C.POINTER(ag_encoder_desc), vp, vp, i32, i32, vp, vp, sz, vp, vp]),
    "ag_reload_knobs": (i32, []),
    "ag_set_stream_cus": (i32, [vp, i32]),
    "ag_seq_compact_plan": (i32, [vp, i32, i32, vp, vp, vp]),
    "ag_gather_rows": (i32, [vp, i64, vp, vp, i64, i32, i32, i32, vp, vp]),
    "ag_masked_attention_varlen": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
}

# test infrastructure (never loaded by the product path): the same library built with -DAG_REF_KERNELS, i.e. with the two earlier
# large-M GEMM generations the shipped stream kernel is checked against bit for bit (tests/test_gpu_gemm_ring.py: use_library)
REF_LIB_PATH = os.path.join(_HERE, "lib", "libautognothi_hip_ref.so")

_lib: Optional[C.CDLL] = None
_handles = {}


def use_library(path: Optional[str] = None) -> None:
    """switch every later call of this process to the library at ``path`` (None: back to LIB_PATH).  Parity tests only."""
    global _lib
    _lib = None if path is None else _load(path)


class HipLibraryMissing(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def _load(path: str) -> C.CDLL:
    if path not in _handles:
        if not os.path.exists(path):
            raise HipLibraryMissing(
                f"{path} is not built; there is no CPU fallback. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` from the repo root.")
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library drift
            fn.restype, fn.argtypes = res, args
        _handles[path] = handle
    return _handles[path]


def check(rc: int) -> None:
    if rc != AG_OK:
        msg = lib().ag_last_error()
        raise RuntimeError(f"autognothi_hip error {rc}: {msg.decode() if msg else '?'}")


def require_gpu(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "autognothi_amd: this op only runs on an MI355X (ROCm) device tensor; got a "
                f"{t.device} tensor. There is deliberately no CPU fallback.")


def ptr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


_BACKGROUND = {}     # device index -> the device's background stream, created AND used at this library's first launch on the device


def background_stream(index: int):
    """The device's second stream for work that runs BESIDE the caller's (the two-stream training epoch: scripts/common.TrainPartition):
    a torch high-priority stream that is created and given one trivial kernel at the library's FIRST launch on the device, and kept.
    Measured (profiles/HISTORY.md §10): a HIP stream takes its hardware queue at its first submission, and a second stream that does
    so after the process has captured hipGraphs / made its other streams runs its kernels behind the caller's instead of beside them
    (the same two-stream schedule: 420 instead of 635 images/s, against 550 on one stream); one that did so first never does."""
    import torch
    st = _BACKGROUND.get(index)
    if st is None:
        dev = torch.device("cuda", index)
        st = _BACKGROUND[index] = torch.cuda.Stream(dev, priority=-1)
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev).add_(1)
        st.synchronize()
    return st


def streams_overlap(a, b, microseconds: int = 150) -> bool:
    """Do torch streams ``a`` and ``b`` (same device) run their kernels BESIDE each other?  One idle wave of ``microseconds`` on each,
    started together: one kernel's wall time = beside, two = behind (HIP multiplexes a process's streams onto a few hardware queues, and
    two streams on one queue serialise: profiles/HISTORY.md §10).  A yes / no observable, ~0.5 ms, synchronises both streams."""
    import time
    import torch
    a.synchronize(); b.synchronize()
    best = 1e9
    with torch.cuda.device(a.device):
        for _ in range(3):
            t0 = time.perf_counter()
            check(lib().ag_probe_spin(microseconds, a.cuda_stream))
            check(lib().ag_probe_spin(microseconds, b.cuda_stream))
            a.synchronize(); b.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e6)
    return best < 1.6 * microseconds


def stream() -> int:
    """raw hipStream_t of torch's current stream on the current device (the C accessors: this runs once per kernel launch)."""
    import torch
    d = torch._C._cuda_getDevice()
    if d not in _BACKGROUND and not torch.cuda.is_current_stream_capturing():   # (never inside a capture: stream creation + synchronize)
        background_stream(d)
    return torch._C._cuda_getCurrentRawStream(d)


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on(device):
    """Device guard for a launch: a no-op object when ``device`` is already current (one process per GPU: always, after
    start-up), ``torch.cuda.device(device)`` otherwise.  torch's own guard costs ~8 us per launch in Python."""
    import torch
    idx = device.index if hasattr(device, "index") else device
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NO_GUARD
    return torch.cuda.device(idx)
