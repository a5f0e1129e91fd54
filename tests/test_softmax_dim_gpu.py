"""GPU parity tests for softmax along any dimension (C ABI myfa_softmax_nd, Python
`softmax(x, dim)`).

The reference's softmax is rows of a 2-D float32 matrix (csrc/api.cpp:110-166,
tests/test_flash_attn.py:132-151); the other dimensions have no counterpart there, so the
oracle `ref_cpu.softmax_dim` is pinned by oracle/make_golden.py against torch.softmax in
float64 and the fixtures tests/golden/softmax_dim_*.npz hold that output.

Tolerances as for the row path: float32 |out - oracle| <= 1e-5; float16 / bfloat16 within one
rounding of the element type of the float64 oracle on the rounded input (2^-11, 2^-8 relative
plus 1e-6).
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

ATOL = 1e-5
LOWP_REL = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}
GOLDEN = ["softmax_dim_300x130_d0", "softmax_dim_4x37x96_d1", "softmax_dim_2x3x50x7_d2",
          "softmax_dim_3x700x5_d1", "softmax_dim_6x33x64_d2"]


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


@pytest.mark.parametrize("name", GOLDEN)
def test_vs_torch_f64_golden(mfa, name):
    g = load_golden(name)
    dim = int(g["dim"])
    out = mfa.softmax(torch.from_numpy(g["x"]).cuda(), dim=dim).cpu().numpy()
    assert np.abs(out - g["out_torch_f64"]).max() <= ATOL
    assert np.allclose(out.sum(axis=dim, dtype=np.float64), 1.0, atol=1e-5)


# every launch class: row segment of 4 / 16 / 64 items, vector and scalar items, column block in
# registers (short and long variant) and streamed, ragged inner sizes, one column block only
SHAPES = [
    ((7, 5, 3), 1), ((2, 9, 4), 1), ((1, 1, 8), 1), ((3, 1, 5), 1), ((2, 40, 12), 1), ((64, 31, 100), 1),
    ((4, 100, 64), 1), ((4, 200, 260), 1), ((2, 1000, 36), 1), ((1, 5000, 16), 1), ((3, 513, 7), 1),
    ((1500, 96), 0), ((33, 4096), 0), ((4096, 33), 0), ((2, 3, 17, 5, 4), 2), ((2, 3, 17, 5, 4), 0),
    ((8, 64, 1024), 1), ((2, 2050, 1024), 1), ((16, 8, 32, 32), 1),
]


@pytest.mark.parametrize("shape,dim", SHAPES)
def test_f32_vs_oracle(mfa, shape, dim):
    torch.manual_seed(1)
    x = torch.randn(*shape) * 4
    out = mfa.softmax(x.cuda(), dim=dim)
    assert out.shape == x.shape and out.dtype == torch.float32
    assert np.abs(out.cpu().numpy() - ref_cpu.softmax_dim(x.numpy(), dim)).max() <= ATOL


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,dim", SHAPES[::2])
def test_lowp_vs_oracle(mfa, shape, dim, dtype):
    torch.manual_seed(2)
    x = (torch.randn(*shape) * 4).to(dtype)
    out = mfa.softmax(x.cuda(), dim=dim)
    assert out.dtype == dtype
    ref = ref_cpu.softmax_dim(x.float().numpy(), dim)
    assert (np.abs(out.float().cpu().numpy() - ref) <= LOWP_REL[dtype] * ref + 1e-6).all()


@pytest.mark.parametrize("dim", [0, 1, 2, -1, -3])
def test_matches_row_path_and_torch(mfa, dim):
    """Every dim of one tensor: against torch.softmax on the GPU, and the last dim bit-for-bit
    against the row entry point it is routed to."""
    torch.manual_seed(3)
    x = torch.randn(12, 50, 72, device="cuda")
    out = mfa.softmax(x, dim=dim)
    assert float((out - torch.softmax(x, dim=dim)).abs().max()) <= ATOL
    if dim in (2, -1):
        assert torch.equal(out, mfa.softmax_rows(x.reshape(-1, 72)).reshape(x.shape))


def test_special_values(mfa):
    """-inf entries are zeros, a column of equal values is uniform, a column that is all -inf is
    NaN (the definition's 0/0, as in torch), +large values do not overflow."""
    x = torch.zeros(2, 6, 8)
    x[0, :, 0] = float("-inf"); x[0, 2, 0] = 1.0          # one finite entry -> one-hot
    x[0, :, 1] = 1e30
    x[1, :, 2] = float("-inf")                             # all -inf -> NaN
    x[1, :, 3] = torch.tensor([0.0, 1.0, 2.0, 3.0, 4.0, 5.0]) + 80000.0
    out = mfa.softmax(x.cuda(), dim=1).cpu()
    ref = torch.softmax(x.double(), dim=1)
    assert torch.isnan(out[1, :, 2]).all() and torch.isnan(ref[1, :, 2]).all()
    mask = ~torch.isnan(ref)
    assert float((out[mask] - ref[mask].float()).abs().max()) <= ATOL
    assert out[0, 2, 0] == 1.0 and float(out[0, :, 0].sum()) == 1.0


def test_large_column_softmax_properties(mfa):
    """8192 x 8192 along dim 0 (streamed column blocks): columns sum to one, sampled columns
    against the oracle, and agreement with the row kernel on the transposed matrix."""
    n = 8192
    torch.manual_seed(4)
    x = torch.randn(n, n, device="cuda")
    out = mfa.softmax(x, dim=0)
    assert float((out.sum(dim=0, dtype=torch.float64) - 1.0).abs().max()) <= 1e-5
    cols = [0, 1, 63, 64, 4097, n - 1]
    ref = ref_cpu.softmax_dim(x[:, cols].cpu().numpy(), 0)
    assert np.abs(out[:, cols].cpu().numpy() - ref).max() <= ATOL
    rows = mfa.softmax_rows(x.t().contiguous())
    assert float((rows.t() - out).abs().max()) <= 2e-7


def test_empty_inplace_and_errors(mfa):
    from my_flash_attn_cuda import _lib
    for shape, dim in [((0, 4, 4), 1), ((4, 0, 4), 1), ((4, 4, 0), 1)]:
        assert mfa.softmax(torch.empty(*shape, device="cuda"), dim=dim).shape == shape
    x = torch.randn(3, 20, 8, device="cuda")
    want = mfa.softmax(x, dim=1)
    rc = _lib.lib.myfa_softmax_nd(x.data_ptr(), x.data_ptr(), 3, 20, 8, 0, None)      # in place
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(x, want)
    assert _lib.lib.myfa_softmax_nd(x.data_ptr(), x.data_ptr(), -1, 20, 8, 0, None) != 0
    assert b"negative" in _lib.lib.myfa_last_error()
    assert _lib.lib.myfa_softmax_nd(x.data_ptr(), x.data_ptr(), 3, 20, 8, 77, None) != 0
    assert _lib.lib.myfa_softmax_nd(None, x.data_ptr(), 3, 20, 8, 0, None) != 0
    with pytest.raises(IndexError):
        mfa.softmax(x, dim=3)
    with pytest.raises(RuntimeError, match="contiguous"):
        mfa.softmax(x.transpose(0, 1), dim=1)
