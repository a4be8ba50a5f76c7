"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dssvae_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dssvae_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dssvae_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "dssvae_gemm_f32" in syms and "dssvae_gru_fwd" in syms and "dssvae_linear_lse_fwd" in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol():
    from dss_vae_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(so, name), "missing export: " + name
    assert sorted(lib.EXPORTS) == declared_symbols(), "lib.py binding table out of sync with the header"
    assert so.dssvae_abi_version() == 9


def test_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback: host tensors are rejected loudly."""
    import torch
    from dss_vae_amd import ops
    x, w = torch.zeros(4, 8), torch.zeros(3, 8)
    with pytest.raises(RuntimeError):
        ops.linear(x, w, None)
    with pytest.raises(RuntimeError):
        ops.linear_ce(x, None, w, None, torch.zeros(4, dtype=torch.long), 0)


def test_product_never_imports_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "dss_vae_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(base, f)


def test_argmax_row_plan_is_host_arithmetic():
    """dssvae_skinny_argmax_padded_rows (no device work: the CU count falls back to 256 without a GPU): full 16-row
    MFMA tiles + one or two left-over rows on the vector ALUs for a single 64-row chunk, else rounded up"""
    from dss_vae_amd import lib
    so = ctypes.CDLL(lib.LIB_PATH)
    f = so.dssvae_skinny_argmax_padded_rows
    f.argtypes, f.restype = [ctypes.c_int, ctypes.c_int], ctypes.c_int
    want = {(50, 30000): 50, (49, 30000): 49, (51, 30000): 64, (40, 30000): 48, (33, 30000): 33, (34, 30000): 34,
            (35, 30000): 48, (64, 30000): 64, (100, 30000): 128, (20, 30000): 32, (50, 300): 64}
    assert {k: f(*k) for k in want} == want
    assert f(0, 5) == -1
