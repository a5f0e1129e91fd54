"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every
symbol include/myfa.h declares, and the Python host layer rejects bad arguments
the way the reference's binding does (csrc/api.cpp, csrc/my_flash_attn/util.hpp:7-9).
No kernel is launched here.
"""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

import my_flash_attn_cuda as mfa
from my_flash_attn_cuda import _lib


def _header_symbols():
    with open(os.path.join(ROOT, "include", "myfa.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(myfa_[a-z0-9_]+)\s*\(", text)))


def test_header_and_loader_agree():
    assert _header_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.myfa_abi_version() == _lib.ABI_VERSION


def test_reference_function_names_present():
    # csrc/api.cpp:194-200
    for name in ("my_softmax", "my_attention", "my_matmul", "my_matmul_cublas", "my_matmul_out"):
        assert callable(getattr(mfa, name))
    assert callable(mfa.flash_attn_fwd)


def test_cpu_tensors_are_rejected_not_computed():
    x = torch.randn(4, 8)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        mfa.my_softmax(x, 1)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        mfa.my_attention(x, x.t().contiguous(), x)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        mfa.my_matmul(x, x.t().contiguous())
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        mfa.flash_attn_fwd(x[None, None], x[None, None], x[None, None])


def test_c_abi_argument_errors_set_last_error():
    lib = _lib.lib
    assert lib.myfa_softmax_f32(None, None, -1, 4, None) == 1
    assert b"negative" in lib.myfa_last_error()
    assert lib.myfa_softmax_f32(None, None, 0, 4, None) == 0          # empty: no-op, no launch
    assert lib.myfa_attn_fwd_f32(None, None, None, None, 0, 8, 1.0, None) == 0
    assert lib.myfa_attn_fwd_f32(None, None, None, None, 4, 8, 1.0, None) == 1
    assert b"null" in lib.myfa_last_error()
    assert lib.myfa_attn_fwd_bhnd(None, None, None, None, None, 1, 1, -2, 64, 1.0, 2, None) == 1
    with pytest.raises(RuntimeError, match="negative"):
        _lib.check(lib.myfa_softmax_f32(None, None, 2, -3, None))
