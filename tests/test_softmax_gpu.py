"""GPU parity tests for the row-softmax path, through the drop-in module and the C ABI.

Mirrors the reference's test_softmax_kernel (tests/test_flash_attn.py:132-151:
(255, 516), my_softmax(x, 5) vs F.softmax, values in [0, 1], rows sum to 1) and
adds the shapes it lists but has commented out (:134-135), ragged and unaligned
widths, every register-resident dispatch class, the long-row streaming kernel,
and the benchmark's sweep sizes (4096^2 ... 32768^2) through size-independent properties.

Tolerance: float32, |out - oracle| <= 1e-5 absolute against the float64 oracle and
against the reference's golden outputs (the reference's own bar is 1e-4).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

ATOL = 1e-5


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


def _check(out, x, oracle_out):
    o = out.cpu().numpy()
    assert (o >= 0).all() and (o <= 1).all()
    assert np.abs(o - oracle_out).max() <= ATOL
    assert np.allclose(o.sum(axis=1, dtype=np.float64), 1.0, atol=1e-5)


@pytest.mark.parametrize("kernel_no", [1, 2, 3, 4, 5])
def test_reference_shape_vs_golden(mfa, kernel_no):
    g = load_golden("softmax_255x516")
    x = torch.from_numpy(g["x"]).cuda()
    out = mfa.my_softmax(x, kernel_no)
    o = out.cpu().numpy()
    assert np.abs(o - g["out_ref"]).max() <= ATOL      # reference's _softmax
    assert np.abs(o - g["out_torch"]).max() <= ATOL    # F.softmax, what its test compares with
    assert (o >= 0).all() and (o <= 1).all()
    assert torch.allclose(out.sum(dim=1), torch.ones(255, device="cuda"))


def test_config1_rows_vs_golden(mfa):
    # the fixture keeps every 8th row of the seeded 1024 x 1024 input; rows are independent
    g = load_golden("softmax_1024x1024")
    o = mfa.my_softmax(torch.from_numpy(g["x"]).cuda(), 5).cpu().numpy()
    assert np.abs(o - g["out_ref"]).max() <= ATOL
    assert np.abs(o - g["out_torch"]).max() <= ATOL


@pytest.mark.parametrize("h,w", [
    (256, 256), (99, 99), (100, 2048), (100, 2047), (1024, 1024),   # reference's commented list
    (1, 1), (3, 1), (1, 5), (7, 4), (5, 63), (5, 64), (5, 65), (300, 252), (300, 256), (300, 260),
    (33, 1000), (33, 1024), (9, 2052), (9, 4096), (9, 4100), (4, 8192), (4, 8196),
    (3, 16384), (3, 16388), (2, 32768), (2, 32764), (2, 32772), (2, 40000), (2, 65536), (3, 8193), (2, 33001),
])
def test_shapes_vs_oracle(mfa, c_oracle, h, w):
    torch.manual_seed(1)
    x = torch.randn(h, w, dtype=torch.float32)
    out = mfa.my_softmax(x.cuda(), 1)
    assert out.shape == (h, w) and out.dtype == torch.float32
    _check(out, x, c_oracle.softmax_rows(x.numpy()))


def test_large_magnitudes_and_shift_invariance(mfa, c_oracle):
    torch.manual_seed(1)
    x = torch.randn(64, 1000, dtype=torch.float32) * 30.0 + 500.0
    out = mfa.my_softmax(x.cuda(), 1)
    _check(out, x, c_oracle.softmax_rows(x.numpy()))
    # softmax(x + c) == softmax(x): exact here because x - max is computed first
    shifted = mfa.my_softmax((x + 64.0).cuda(), 1)
    ref = c_oracle.softmax_rows((x + 64.0).numpy())
    assert np.abs(shifted.cpu().numpy() - ref).max() <= ATOL


def test_unaligned_view_base(mfa, c_oracle):
    # a contiguous tensor whose storage offset is not 16-byte aligned takes the scalar path
    torch.manual_seed(1)
    buf = torch.randn(1 + 16 * 64, dtype=torch.float32).cuda()
    x = buf[1:].view(16, 64)
    assert x.is_contiguous() and x.data_ptr() % 16 != 0
    _check(mfa.my_softmax(x, 1), x.cpu(), c_oracle.softmax_rows(x.cpu().numpy()))


def test_errors(mfa):
    x = torch.randn(4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported kernel number"):
        mfa.my_softmax(x, 6)
    with pytest.raises(RuntimeError, match="contiguous"):
        mfa.my_softmax(x.t(), 1)
    with pytest.raises(RuntimeError, match="Float"):
        mfa.my_softmax(x.half(), 1)
    assert mfa.my_softmax(torch.empty(0, 8, device="cuda"), 1).shape == (0, 8)


def test_nondefault_stream(mfa, c_oracle):
    torch.manual_seed(1)
    x = torch.randn(128, 516).cuda()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = mfa.my_softmax(x, 1)
    s.synchronize()
    _check(out, x.cpu(), c_oracle.softmax_rows(x.cpu().numpy()))


@pytest.mark.parametrize("n", [4096, 8192, 16384, 32768])
def test_sweep_sizes_properties(mfa, c_oracle, n):
    """The benchmark's n x n sweep: properties over the whole matrix, oracle on sampled rows."""
    torch.manual_seed(1)
    x = torch.randn(n, n, dtype=torch.float32, device="cuda")
    out = mfa.my_softmax(x, 1)
    assert bool((out >= 0).all()) and bool((out <= 1).all())
    sums = out.sum(dim=1, dtype=torch.float64)
    assert float((sums - 1).abs().max()) < 1e-5
    # argmax is preserved, and out * sum(exp(x - max)) == exp(x - max)
    assert bool((out.argmax(dim=1) == x.argmax(dim=1)).all())
    rows = torch.tensor([0, 1, n // 3, n // 2 + 1, n - 2, n - 1], device="cuda")
    ref = c_oracle.softmax_rows(x[rows].cpu().numpy())
    assert np.abs(out[rows].cpu().numpy() - ref).max() <= ATOL
    # idempotence of the normalisation: softmax(log(out)) == out
    again = mfa.my_softmax(torch.log(out[rows].contiguous()), 1)
    assert float((again - out[rows]).abs().max()) <= ATOL


# ---- float16 / bfloat16 rows (softmax_rows; the reference's my_softmax is float32 only) ----
# Bound: the float32 result rounded once to the output type -- half an ulp of a value <= 1,
# 2^-12 for float16 and 2^-9 for bfloat16 -- plus the float32 kernel's own 1e-5.
LOWP_ATOL = {torch.float16: 2.0 ** -12 + 1e-5, torch.bfloat16: 2.0 ** -9 + 1e-5}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("h,w", [(1, 1), (3, 7), (255, 516), (64, 8), (33, 1000), (16, 4096), (7, 8192),
                                 (5, 16384), (3, 32768), (2, 65536), (2, 70001), (4, 12345)])
def test_lowp_rows_vs_oracle(mfa, c_oracle, dtype, h, w):
    torch.manual_seed(1)
    x = (torch.randn(h, w) * 3).to(dtype)
    out = mfa.softmax_rows(x.cuda())
    assert out.dtype == dtype and out.shape == (h, w)
    ref = c_oracle.softmax_rows(x.float().numpy())
    o = out.float().cpu().numpy()
    assert (o >= 0).all() and (o <= 1).all()
    assert np.abs(o - ref).max() <= LOWP_ATOL[dtype]


def test_lowp_matches_float32_kernel_on_the_same_values(mfa):
    """Same arithmetic whatever the element type: the 16-bit result is the float32 result of the
    widened input rounded once -- up to the order of the row sum, which follows the launch
    shape (items are 16 bytes, so 8 halves against 4 floats), i.e. up to one ulp of the output."""
    torch.manual_seed(2)
    for dtype in (torch.float16, torch.bfloat16):
        x = (torch.randn(37, 2048) * 2).to(dtype).cuda()
        a = mfa.softmax_rows(x)
        b = mfa.my_softmax(x.float(), 1).to(dtype)
        assert float((a.float() - b.float()).abs().max()) <= 2 * LOWP_ATOL[dtype]
        assert float((a != b).float().mean()) < 0.01
    xf = torch.randn(9, 300, device="cuda")
    assert torch.equal(mfa.softmax_rows(xf), mfa.my_softmax(xf, 1))


def test_lowp_errors(mfa):
    with pytest.raises(RuntimeError, match="unsupported dtype"):
        mfa.softmax_rows(torch.zeros(2, 2, dtype=torch.float64, device="cuda"))
    with pytest.raises(RuntimeError, match="2-D"):
        mfa.softmax_rows(torch.zeros(2, 2, 2, dtype=torch.bfloat16, device="cuda"))


# rows that do not start on a 16-byte boundary (w not a multiple of 4 floats / 8 halves) are read as
# the aligned 16-byte groups that cover them, first and last group element by element; every
# launch class of that path, the streamed one (w > 32768) included, and its edges
UNALIGNED = [(7, 9), (5, 33), (64, 255), (33, 257), (100, 1023), (9, 1025), (6, 2049), (5, 4097), (3, 8191),
             (3, 12345), (2, 32767), (3, 40001), (2, 70003)]


@pytest.mark.parametrize("shape", UNALIGNED)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_unaligned_rows_vs_oracle(mfa, shape, dtype):
    from oracle import ref_cpu
    torch.manual_seed(21)
    h, w = shape
    x = (torch.randn(h, w) * 3).to(dtype)
    x[0, : min(w - 2, 5000)] = float("-inf")          # leading -inf (left padding) in the first row
    guard = torch.full((h * w + 64,), 7.0, dtype=dtype, device="cuda")   # the output sits inside a guarded buffer
    out = guard[32:32 + h * w].view(h, w)
    xd = torch.empty(h * w + 64, dtype=dtype, device="cuda")[32:32 + h * w].view(h, w)
    xd.copy_(x)
    from my_flash_attn_cuda import _lib
    tag = {torch.float32: _lib.MYFA_F32, torch.float16: _lib.MYFA_F16, torch.bfloat16: _lib.MYFA_BF16}[dtype]
    rc = _lib.lib.myfa_softmax(xd.data_ptr(), out.data_ptr(), h, w, tag, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0
    ref = ref_cpu.softmax_dim(x.float().numpy(), 1)
    tol = {torch.float32: 0.0, torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype]
    assert (np.abs(out.float().cpu().numpy() - ref) <= tol * ref + 1e-5).all()
    # nothing outside the rows was written
    assert bool((guard[:32] == 7.0).all()) and bool((guard[32 + h * w:] == 7.0).all())
    # the same values as the element-wise path (tensor base off the 16-byte grid)
    off = torch.empty(h * w + 64, dtype=dtype, device="cuda")
    xo, oo = off[1:1 + h * w].view(h, w), torch.empty(h * w + 64, dtype=dtype, device="cuda")[1:1 + h * w].view(h, w)
    xo.copy_(x)
    rc = _lib.lib.myfa_softmax(xo.data_ptr(), oo.data_ptr(), h, w, tag, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0
    assert float((oo.float() - out.float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 2e-3)
