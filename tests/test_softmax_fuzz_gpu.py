"""Randomised GPU parity sweep of the softmax family: shapes drawn around the launch thresholds
(one wave per row, several waves per row, register-resident up to 65536 floats, streamed, rows off
the 16-byte grid, few long rows with the workspace split, register / streamed / row-split column
blocks), every dtype, with and without the fused prologue, against torch.softmax in float64.

The reference's softmax is rows of a 2-D float32 matrix (csrc/api.cpp:110-166); everything else here
has no counterpart there and is pinned through `oracle.ref_cpu.softmax_dim` (tests/test_oracle.py)
to the same torch.softmax in float64 that this sweep calls directly on the GPU.
Tolerances: float32 1e-5 absolute; 16-bit one rounding of the element type (relative) plus 1e-6.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL = 1e-5
LOWP_REL = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}
EDGE_W = [1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1023, 1025, 2047, 2049, 4097, 8191, 8193, 16385,
          32767, 32769, 50257, 65535, 65536, 65537, 65544, 70001, 131073]


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


def _row_cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        w = int(rng.choice(EDGE_W)) if rng.random() < 0.6 else int(rng.integers(1, 90000))
        h = int(rng.integers(1, max(2, min(300, (1 << 23) // w))))
        out.append((h, w, int(rng.integers(0, 3)), int(rng.integers(0, 4)), int(rng.integers(0, 1 << 30))))
    return out


def _col_cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        ln = int(rng.choice([1, 2, 17, 64, 65, 255, 1024, 1025, 2049, 4100, 9000]))
        inner = int(rng.choice([2, 3, 4, 8, 12, 31, 32, 36, 64, 100, 256, 260, 1024]))
        outer = int(rng.integers(1, max(2, min(6, (1 << 22) // (ln * inner)))))
        out.append((outer, ln, inner, int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 1 << 30))))
    return out


def _check(out, ref, dtype):
    ok = ~torch.isnan(ref)
    assert bool((torch.isnan(out.double()) == torch.isnan(ref)).all())
    if dtype == torch.float32:
        assert float((out.double()[ok] - ref[ok]).abs().max()) <= ATOL
    else:
        assert bool(((out.double()[ok] - ref[ok]).abs() <= LOWP_REL[dtype] * ref[ok] + 1e-6).all())


@pytest.mark.parametrize("case", _row_cases(31, 60))
def test_fuzz_rows(mfa, case):
    h, w, dt, mode, seed = case
    dtype = [torch.float32, torch.float16, torch.bfloat16][dt]
    torch.manual_seed(seed)
    x = (torch.randn(h, w, device="cuda") * 3).to(dtype)
    if w > 4:
        x[0, : w // 3] = float("-inf")                  # left padding in the first row
    if mode == 0:                                        # plain: every Python entry point agrees
        out = mfa.softmax(x, dim=-1)
        assert torch.equal(torch.nan_to_num(mfa.softmax_rows(x)), torch.nan_to_num(out))
        if dtype == torch.float32:
            assert torch.equal(torch.nan_to_num(mfa.my_softmax(x, 1)), torch.nan_to_num(out))
        ref = torch.softmax(x.double(), -1)
    elif mode == 1:                                      # scale
        out = mfa.softmax(x, dim=-1, scale=0.37)
        ref = torch.softmax(0.37 * x.double(), -1)
    elif mode == 2:                                      # scale + a mask row shared by all rows
        mask = torch.randn(w, device="cuda")
        mask[torch.rand(w, device="cuda") < 0.2] = float("-inf")
        out = mfa.softmax(x, dim=-1, scale=1.7, mask=mask)
        ref = torch.softmax(1.7 * x.double() + mask.double(), -1)
    else:                                                # a mask of the tensor's own shape
        mask = torch.randn(h, w, device="cuda")
        out = mfa.softmax(x, dim=-1, mask=mask)
        ref = torch.softmax(x.double() + mask.double(), -1)
    assert out.dtype == dtype and out.shape == x.shape
    _check(out, ref, dtype)


@pytest.mark.parametrize("case", _col_cases(32, 60))
def test_fuzz_columns(mfa, case):
    outer, ln, inner, dt, mode, seed = case
    dtype = [torch.float32, torch.float16, torch.bfloat16][dt]
    torch.manual_seed(seed)
    x = (torch.randn(outer, ln, inner, device="cuda") * 3).to(dtype)
    if ln > 4:
        x[0, : ln // 2, 0] = float("-inf")              # leading -inf along the reduced dimension
        x[0, :, inner - 1] = float("-inf")              # a slice that is -inf throughout: NaN, as in torch
    if mode == 0:
        out = mfa.softmax(x, dim=1)
        ref = torch.softmax(x.double(), 1)
    elif mode == 1:
        out = mfa.softmax(x, dim=1, scale=0.6)
        ref = torch.softmax(0.6 * x.double(), 1)
    else:
        mask = torch.randn(ln, inner, device="cuda")
        out = mfa.softmax(x, dim=1, scale=1.3, mask=mask)
        ref = torch.softmax(1.3 * x.double() + mask.double(), 1)
    _check(out, ref, dtype)


# the 1024-thread register-resident column kernel: 256-byte row segments up to 1024 rows, 128-byte
# ones up to 2048, ragged last column block, with and without the prologue
@pytest.mark.parametrize("shape,dim", [((16, 1000, 1100), 1), ((8, 1500, 2048), 1), ((2048, 8192), 0), ((300, 1024, 64), 1),
                                       ((2, 4, 1025, 1024), 2)])
def test_long_register_columns(mfa, shape, dim):
    torch.manual_seed(33)
    x = torch.randn(*shape, device="cuda") * 3
    idx = [0] * len(shape)
    idx[dim] = slice(0, shape[dim] // 2)
    x[tuple(idx)] = float("-inf")                        # leading -inf along the reduced dimension of one slice
    _check(mfa.softmax(x, dim=dim), torch.softmax(x.double(), dim), torch.float32)
    mask = torch.randn(*shape[dim:], device="cuda")
    _check(mfa.softmax(x, dim=dim, scale=0.5, mask=mask), torch.softmax(0.5 * x.double() + mask.double(), dim), torch.float32)
    if shape[dim] <= 1024:                               # 16-bit elements: half the rows
        xh = x.to(torch.bfloat16)
        _check(mfa.softmax(xh, dim=dim), torch.softmax(xh.double(), dim), torch.bfloat16)
