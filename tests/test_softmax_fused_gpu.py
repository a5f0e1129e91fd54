"""GPU parity tests for the softmax with the fused scale / mask prologue (C ABI
myfa_softmax_nd_ex, Python `softmax(x, dim, scale=, mask=, causal=)`), and for -inf inputs
on every softmax kernel.

The reference composes attention as matmul -> softmax -> matmul (csrc/api.cpp:186-187) with no
scale and no mask; this entry point is what a caller keeping that composition on MI355X needs.
It has no counterpart in the reference, so the oracle `ref_cpu.softmax_dim(scale=, mask=,
causal=)` is pinned by oracle/make_golden.py against torch.softmax(scale * x + mask) in float64
(fixtures tests/golden/softmax_fused_*.npz).

Tolerances as for the plain softmax: float32 |out - oracle| <= 1e-5; float16 / bfloat16 within
one rounding of the element type of the float64 oracle on the rounded input.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

ATOL = 1e-5
LOWP_REL = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}
GOLDEN = ["softmax_fused_0_2x3x40x40", "softmax_fused_1_2x2x33x33", "softmax_fused_2_3x50x50",
          "softmax_fused_3_4x37x96", "softmax_fused_4_2x300x20", "softmax_fused_5_5x64"]


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


@pytest.mark.parametrize("name", GOLDEN)
def test_vs_torch_f64_golden(mfa, name):
    g = load_golden(name)
    mask = torch.from_numpy(g["mask"]).cuda() if bool(g["has_mask"]) else None
    out = mfa.softmax(torch.from_numpy(g["x"]).cuda(), dim=int(g["dim"]), scale=float(g["scale"]),
                      mask=mask, causal=bool(g["causal"])).cpu().numpy()
    assert np.isfinite(out).all()
    assert np.abs(out - g["out_torch_f64"]).max() <= ATOL


def _mask(shape, seed, frac=0.25):
    gen = torch.Generator().manual_seed(seed)
    m = torch.randn(*shape, generator=gen)
    m[torch.rand(*shape, generator=gen) < frac] = float("-inf")
    return m


# (shape, dim, mask shape, causal): every kernel class of the softmax family behind the prologue --
# one wave per row, several waves per row, streamed rows (> 32768 and, unaligned, > 8192),
# register column blocks and streamed columns, vector and scalar items
CASES = [
    ((3, 200, 200), -1, (200, 200), False), ((2, 2, 129, 129), -1, None, True), ((2, 515, 515), -1, (515, 515), True),
    ((4, 4096), -1, (4096,), False), ((2, 40000), -1, (2, 40000), False), ((3, 10001), -1, (10001,), False),
    ((3, 40003), -1, (40003,), False), ((5, 1023, 1023), -1, None, True),
    ((2, 1024, 1024), -1, (1024, 1024), True), ((4, 100, 64), 1, (100, 64), False), ((2, 4, 300, 20), 2, (4, 300, 20), False),
    ((2, 2050, 1024), 1, (2050, 1024), False), ((4096, 512), 0, (4096, 512), False), ((1500, 97), 0, (1500, 97), False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,dim,mshape,causal", CASES)
def test_fused_vs_oracle(mfa, shape, dim, mshape, causal, dtype):
    torch.manual_seed(3)
    x = (torch.randn(*shape) * 3).to(dtype)
    scale = 0.37
    mask = None
    if mshape is not None:
        mask = _mask(mshape, 4)
        # one finite entry per slice along the softmax axis, so that no slice is -inf throughout
        md = dim if dim < 0 else dim - len(shape)
        idx = [slice(None)] * len(mshape)
        idx[md] = 0
        mask[tuple(idx)] = 0.5
    out = mfa.softmax(x.cuda(), dim=dim, scale=scale, mask=None if mask is None else mask.cuda(), causal=causal)
    assert out.dtype == dtype and out.shape == x.shape
    ref = ref_cpu.softmax_dim(x.float().numpy(), dim, scale=scale, mask=None if mask is None else mask.numpy(), causal=causal)
    got = out.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    if dtype == torch.float32:
        assert np.abs(got - ref).max() <= ATOL
    else:
        assert (np.abs(got - ref) <= LOWP_REL[dtype] * np.abs(ref) + 1e-6).all()


def test_neutral_prologue_equals_plain_softmax(mfa):
    """scale = 1, no mask, no flag through the C ABI of the fused entry point: bit-identical to
    myfa_softmax_nd for rows and for columns."""
    from my_flash_attn_cuda import _lib
    for shape, outer, ln, inner in (((64, 300), 64, 300, 1), ((8, 100, 64), 8, 100, 64)):
        torch.manual_seed(5)
        x = torch.randn(*shape, device="cuda")
        a, b = torch.empty_like(x), torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib.myfa_softmax_nd(x.data_ptr(), a.data_ptr(), outer, ln, inner, _lib.MYFA_F32, st))
        _lib.check(_lib.lib.myfa_softmax_nd_ex(x.data_ptr(), b.data_ptr(), outer, ln, inner, _lib.MYFA_F32, 1.0, None, 1, 0, st))
        assert torch.equal(a, b)


@pytest.mark.parametrize("w", [300, 5000, 40000, 10001])
def test_rows_with_leading_minus_inf(mfa, w):
    """-inf is ordinary input (left padding, sliding windows): rows that BEGIN with -inf, rows with
    -inf everywhere but one place, in every row kernel (registers, streamed with 16-byte items at
    40000, streamed element-wise at 10001).  A row that is -inf throughout is NaN, as in torch."""
    torch.manual_seed(6)
    x = torch.randn(6, w)
    x[0, : w // 2] = float("-inf")
    x[1, : w - 1] = float("-inf")
    x[2, 1:] = float("-inf")
    x[3, ::2] = float("-inf")
    x[4, :] = float("-inf")
    x[5, : min(2048, w - 3)] = float("-inf")
    out = mfa.softmax(x.cuda(), dim=1).cpu()
    ref = torch.softmax(x.double(), dim=1)
    assert bool(torch.isnan(out[4]).all())
    keep = [0, 1, 2, 3, 5]
    assert bool(torch.isfinite(out[keep]).all())
    assert float((out[keep].double() - ref[keep]).abs().max()) <= ATOL


@pytest.mark.parametrize("shape", [(4096, 512), (6, 64), (3000, 36)])
def test_columns_with_leading_minus_inf(mfa, shape):
    """The same along dim 0 (streamed columns at (4096, 512) and (3000, 36), registers at (6, 64)):
    a causal mask along this dimension gives columns whose leading rows are -inf; one column that is
    -inf throughout must still be NaN."""
    torch.manual_seed(7)
    L, I = shape
    x = torch.randn(L, I)
    x = x.masked_fill(torch.arange(L)[:, None] < (torch.arange(I)[None, :] * (L - 1) // max(I - 1, 1)), float("-inf"))
    x[:, 3] = float("-inf")
    out = mfa.softmax(x.cuda(), dim=0).cpu()
    ref = torch.softmax(x.double(), dim=0)
    cols = [c for c in range(I) if c != 3]
    assert bool(torch.isnan(out[:, 3]).all())
    assert bool(torch.isfinite(out[:, cols]).all())
    assert float((out[:, cols].double() - ref[:, cols]).abs().max()) <= ATOL


def test_fused_errors(mfa):
    x = torch.randn(2, 8, 8, device="cuda")
    with pytest.raises(RuntimeError, match="causal"):
        mfa.softmax(x, dim=1, causal=True)
    with pytest.raises(RuntimeError, match="causal"):
        mfa.softmax(torch.randn(2, 8, 9, device="cuda"), dim=-1, causal=True)
    with pytest.raises(RuntimeError, match="mask shape"):
        mfa.softmax(x, dim=-1, mask=torch.zeros(7, 8, device="cuda"))
    with pytest.raises(RuntimeError, match="mask shape"):
        mfa.softmax(x, dim=1, mask=torch.zeros(8, device="cuda"))     # does not cover dim
    with pytest.raises(RuntimeError, match="Float"):
        mfa.softmax(x, dim=-1, mask=torch.zeros(8, 8, device="cuda", dtype=torch.float16))
