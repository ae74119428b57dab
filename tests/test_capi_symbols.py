"""CPU: the C-ABI shared library loads and exports every symbol include/autognothi_hip.h declares, and the
ctypes table in autognothi_amd/_lib.py covers exactly that set (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "autognothi_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ag_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ("ag_mask_shapley_new", "ag_gemm", "ag_masked_attention", "ag_layernorm", "ag_encoder_forward",
                 "ag_shapley_normalize", "ag_shapley_loss", "ag_kl_loss", "ag_perturbed_masks", "ag_mt19937_seed"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from autognothi_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build() first")
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.lib().ag_abi_version() == 6


def test_product_path_has_no_cpu_fallback():
    import torch
    from autognothi_amd import _lib as L, ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pack_mask(torch.ones((2, 196), dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-5, L.AG_F32)


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "autognothi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports oracle"
