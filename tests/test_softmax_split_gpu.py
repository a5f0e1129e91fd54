"""GPU parity tests for the column softmax with a caller workspace (C ABI myfa_softmax_nd_ws /
myfa_softmax_nd_ws_bytes; Python `softmax(x, dim)` takes this route by itself): long columns whose
column blocks alone are too few workgroups for the chip get their rows split over workgroups.

No counterpart in the reference (its softmax is rows of a 2-D float32 matrix, csrc/api.cpp:110-166);
the oracle is `ref_cpu.softmax_dim`, pinned against torch.softmax in float64 by oracle/make_golden.py.
Tolerances as for the other softmax paths: float32 |out - oracle| <= 1e-5; 16-bit within one rounding
of the element type.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu

pytestmark = pytest.mark.gpu

ATOL = 1e-5
LOWP_REL = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}
# (shape, dim): tall and narrow, ragged slices, several outer indices, rows of 4 / 8 / 16+ items
SHAPES = [((5000, 16), 0), ((2, 3000, 36), 1), ((8192, 256), 0), ((4100, 1024), 0), ((100001, 24), 0),
          ((3, 2050, 1024), 1), ((2100, 1000), 0), ((129, 8), 0)]


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


def _view(shape, dim):
    outer = int(np.prod(shape[:dim], dtype=np.int64))
    inner = int(np.prod(shape[dim + 1:], dtype=np.int64))
    return outer, shape[dim], inner


def _call_ws(x, dim, ws, nbytes, out=None, scale=1.0, mask=None, mask_outer=1):
    from my_flash_attn_cuda import _lib
    tag = {torch.float32: _lib.MYFA_F32, torch.float16: _lib.MYFA_F16, torch.bfloat16: _lib.MYFA_BF16}[x.dtype]
    outer, ln, inner = _view(tuple(x.shape), dim)
    out = torch.empty_like(x) if out is None else out
    rc = _lib.lib.myfa_softmax_nd_ws(x.data_ptr(), out.data_ptr(), outer, ln, inner, tag, scale,
                                     mask.data_ptr() if mask is not None else None, mask_outer, 0,
                                     ws.data_ptr() if ws is not None else None, nbytes,
                                     torch.cuda.current_stream().cuda_stream)
    return rc, out


def _ws_bytes(x, dim):
    from my_flash_attn_cuda import _lib
    tag = {torch.float32: _lib.MYFA_F32, torch.float16: _lib.MYFA_F16, torch.bfloat16: _lib.MYFA_BF16}[x.dtype]
    return _lib.lib.myfa_softmax_nd_ws_bytes(*_view(tuple(x.shape), dim), tag)


@pytest.mark.parametrize("shape,dim", SHAPES)
def test_f32_split_vs_oracle_and_unsplit(mfa, shape, dim):
    torch.manual_seed(11)
    x = (torch.randn(*shape) * 4).cuda()
    nbytes = _ws_bytes(x, dim)
    if shape == (129, 8):
        assert nbytes == 0                       # fits one workgroup's registers: nothing to split
        return
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    rc, out = _call_ws(x, dim, ws, nbytes)
    assert rc == 0
    rc, plain = _call_ws(x, dim, None, 0)        # no workspace: the one-launch streamed form
    assert rc == 0
    assert float((out - plain).abs().max()) <= 1e-6
    # the oracle on sampled slices along the other dimensions (all of the tensor for the small ones)
    xn, on = x.cpu().numpy(), out.cpu().numpy()
    if xn.size <= 1 << 22:
        assert np.abs(on - ref_cpu.softmax_dim(xn, dim)).max() <= ATOL
    else:
        xm, om = np.moveaxis(xn, dim, 0).reshape(shape[dim], -1), np.moveaxis(on, dim, 0).reshape(shape[dim], -1)
        cols = [0, 1, 63, 64, om.shape[1] // 2 + 1, om.shape[1] - 1]
        assert np.abs(om[:, cols] - ref_cpu.softmax_dim(xm[:, cols], 0)).max() <= ATOL
    assert float((out.sum(dim=dim, dtype=torch.float64) - 1.0).abs().max()) <= 1e-5
    # the same workspace again, dirty from the first call, and the Python entry point (which takes this route)
    rc, again = _call_ws(x, dim, ws, nbytes)
    assert rc == 0 and torch.equal(again, out)
    assert torch.equal(mfa.softmax(x, dim=dim), out)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,dim", [((5000, 32), 0), ((2, 3000, 72), 1), ((8192, 512), 0)])
def test_lowp_split_vs_oracle(mfa, shape, dim, dtype):
    torch.manual_seed(12)
    x = (torch.randn(*shape) * 4).to(dtype).cuda()
    assert _ws_bytes(x, dim) > 0
    out = mfa.softmax(x, dim=dim)
    ref = ref_cpu.softmax_dim(x.float().cpu().numpy(), dim)
    assert (np.abs(out.float().cpu().numpy() - ref) <= LOWP_REL[dtype] * ref + 1e-6).all()


def test_split_special_values_mask_and_inplace(mfa):
    """Leading -inf rows (a causal mask along the reduced dimension) and slices that are -inf
    throughout are finite, a column that is -inf throughout is NaN as in torch; scale and an additive
    mask are applied in both launches; in place is allowed."""
    torch.manual_seed(13)
    n, w = 6000, 64
    x = torch.randn(n, w) * 3
    x[:4000, 0] = float("-inf")                 # whole slices of -inf, finite tail
    x[:, 1] = float("-inf")                     # NaN column
    x[100:5900, 2] = float("-inf")              # finite head and tail only
    x[:, 3] = 7.0                               # uniform
    xd = x.cuda()
    nbytes = _ws_bytes(xd, 0)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    rc, out = _call_ws(xd, 0, ws, nbytes)
    assert rc == 0
    ref = torch.softmax(x.double(), 0)
    assert torch.isnan(out[:, 1]).all() and torch.isnan(ref[:, 1]).all()
    ok = ~torch.isnan(ref)
    assert float((out.cpu()[ok] - ref[ok].float()).abs().max()) <= ATOL
    assert float(out[:4000, 0].abs().max()) == 0.0

    mask = (torch.randn(n, w) * 2)
    mask[:3000, 5] = float("-inf")
    md = mask.cuda()
    rc, out = _call_ws(xd, 0, ws, nbytes, scale=0.37, mask=md, mask_outer=1)
    assert rc == 0
    ref = torch.softmax(0.37 * x.double() + mask.double(), 0)
    ok = ~torch.isnan(ref)
    assert float((out.cpu()[ok] - ref[ok].float()).abs().max()) <= ATOL
    assert torch.equal(torch.nan_to_num(mfa.softmax(xd, dim=0, scale=0.37, mask=md)), torch.nan_to_num(out))   # (NaN column)

    y = torch.randn(n, w, device="cuda")
    rc, want = _call_ws(y, 0, ws, nbytes)
    rc2, _ = _call_ws(y, 0, ws, nbytes, out=y)
    torch.cuda.synchronize()
    assert rc == 0 and rc2 == 0 and torch.equal(y, want)


def test_workspace_argument_checks(mfa):
    from my_flash_attn_cuda import _lib
    x = torch.randn(5000, 64, device="cuda")
    nbytes = _ws_bytes(x, 0)
    ws = torch.empty(nbytes + 16, dtype=torch.uint8, device="cuda")
    rc, want = _call_ws(x, 0, None, 0)
    assert rc == 0
    rc, small = _call_ws(x, 0, ws, nbytes - 1)   # too small: ignored, the unsplit form runs
    assert rc == 0 and torch.equal(small, want)
    rc, _ = _call_ws(x, 0, ws[4:], nbytes)       # not 16-byte aligned
    assert rc != 0 and b"workspace" in _lib.lib.myfa_last_error()
    rc, _ = _call_ws(x, 0, None, 64)             # a size without a pointer
    assert rc != 0
    rc, _ = _call_ws(x, 0, ws, -1)
    assert rc != 0
    # shapes that never split: rows, short columns, empty tensors
    assert _lib.lib.myfa_softmax_nd_ws_bytes(4096, 4096, 1, _lib.MYFA_F32) == 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(64, 31, 100, _lib.MYFA_F32) == 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(1, 2000, 8192, _lib.MYFA_F32) == 0     # 1024 threads hold 2048 rows in registers
    assert _lib.lib.myfa_softmax_nd_ws_bytes(0, 5000, 64, _lib.MYFA_F32) == 0
    # float32 rows wide enough for a workgroup per CU on 128-byte segments stream in one launch; 16-bit ones split
    assert _lib.lib.myfa_softmax_nd_ws_bytes(1, 2100, 8192, _lib.MYFA_F32) == 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(1, 2100, 8192, _lib.MYFA_BF16) > 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(1, 5000, 64, 77) == 0


# few long rows (inner == 1): each row split over workgroups.  Aligned and off-grid widths, one row,
# widths that leave a short last slice
ROW_SHAPES = [(1, 1 << 20), (4, 300001), (16, 131072), (3, 40961), (100, 70001), (255, 66000)]


@pytest.mark.parametrize("shape", ROW_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_few_long_rows_split(mfa, shape, dtype):
    torch.manual_seed(14)
    h, w = shape
    x = (torch.randn(h, w) * 3).to(dtype)
    x[0, : w // 2] = float("-inf")                 # whole slices of -inf in the first row
    xd = x.cuda()
    nbytes = _ws_bytes(xd, 1)
    # (rows up to 65536 wide fit a workgroup's registers and are not split)
    assert (nbytes > 0) == (w > 65536)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    rc, out = _call_ws(xd, 1, ws, nbytes)
    assert rc == 0
    rc, plain = _call_ws(xd, 1, None, 0)           # no workspace: one workgroup per row
    assert rc == 0
    assert float((out.float() - plain.float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 1e-3)
    ref = ref_cpu.softmax_dim(x.float().numpy(), 1)
    tol = 0.0 if dtype == torch.float32 else LOWP_REL[dtype]
    assert (np.abs(out.float().cpu().numpy() - ref) <= tol * ref + ATOL).all()
    assert float(out[0, : w // 2].float().abs().max()) == 0.0
    rc, again = _call_ws(xd, 1, ws, nbytes)        # the same workspace again, dirty
    assert rc == 0 and torch.equal(again, out)
    assert torch.equal(mfa.softmax(xd, dim=-1), out)
    assert torch.equal(mfa.softmax_rows(xd), out)
    if dtype == torch.float32:
        assert torch.equal(mfa.my_softmax(xd, 5), out)


def test_few_long_rows_fused_and_unsplit_shapes(mfa):
    from my_flash_attn_cuda import _lib
    torch.manual_seed(15)
    h, w = 3, 70001
    x, mask = torch.randn(h, w) * 2, torch.randn(w)
    mask[:30000] = float("-inf")
    out = mfa.softmax(x.cuda(), dim=-1, scale=0.7, mask=mask.cuda())
    ref = torch.softmax(0.7 * x.double() + mask.double(), -1)
    assert float((out.cpu() - ref.float()).abs().max()) <= ATOL
    # rows enough for the chip, rows a workgroup holds in registers: not split
    assert _lib.lib.myfa_softmax_nd_ws_bytes(256, 1 << 20, 1, _lib.MYFA_F32) == 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(8, 65536, 1, _lib.MYFA_F32) == 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(8, 65540, 1, _lib.MYFA_F32) > 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(8, 65536, 1, _lib.MYFA_BF16) == 0
    assert _lib.lib.myfa_softmax_nd_ws_bytes(8, 65544, 1, _lib.MYFA_BF16) > 0
