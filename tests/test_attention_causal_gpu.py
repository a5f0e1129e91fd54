"""GPU parity tests for the causal mask of the batched attention forward (C ABI flag
MYFA_ATTN_CAUSAL, Python `flash_attn_fwd(..., causal=True)`).

The reference has no masked attention (tests/test_flash_attn.py:31-101 and csrc/api.cpp:168-192
are unmasked), so the oracle here is pinned against PyTorch's own
scaled_dot_product_attention(is_causal=True) in float64: `oracle/make_golden.py` checks both
oracles against it and stores its outputs in tests/golden/attn_causal_*.npz.

Tolerances as for the unmasked path: float32 <= 1e-5 and f16/bf16 <= 2e-2 against the float64
oracle (on the rounded inputs for the low-precision types).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

F32_ATOL = 1e-5
LP_ATOL = 2e-2
GOLDEN = ["attn_causal_1x2x100x64", "attn_causal_2x1x257x128", "attn_causal_1x1x64x40"]


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


def _maxdiff(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def _rounded(x, dtype):
    return x.to(dtype).to(torch.float32).numpy()


@pytest.mark.parametrize("name", GOLDEN)
def test_causal_f32_vs_sdpa_golden(mfa, name):
    g = load_golden(name)
    q, k, v = (torch.from_numpy(g[x]).cuda() for x in ("q", "k", "v"))
    out = mfa.flash_attn_fwd(q, k, v, scale=float(g["scale"]), causal=True)
    assert _maxdiff(out.cpu().numpy(), g["out_sdpa_f64"]) <= F32_ATOL


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", GOLDEN[:2])
def test_causal_lowp_vs_sdpa_golden(mfa, name, dtype):
    """Low precision on the golden inputs, judged against the oracle on the ROUNDED inputs
    (at scale 1.0 the rounding of the inputs alone moves the result by more than the bound)."""
    g = load_golden(name)
    q, k, v = (torch.from_numpy(g[x]).cuda().to(dtype) for x in ("q", "k", "v"))
    out = mfa.flash_attn_fwd(q, k, v, scale=float(g["scale"]), causal=True)
    qr, kr, vr = (_rounded(torch.from_numpy(g[x]), dtype) for x in ("q", "k", "v"))
    ref = ref_cpu.attention_bhnd(qr, kr, vr, scale=float(g["scale"]), causal=True)
    assert _maxdiff(out.float().cpu().numpy(), ref) <= LP_ATOL


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [
    (1, 1, 1, 64), (1, 2, 63, 128), (2, 3, 200, 64), (1, 2, 256, 128), (1, 2, 257, 128),
    (1, 3, 1025, 64), (2, 2, 1000, 128), (1, 1, 2048, 128),
])
@pytest.mark.parametrize("scaled", [False, True])
def test_causal_lowp_vs_oracle(mfa, dtype, shape, scaled):
    B, H, N, d = shape
    scale = d ** -0.5 if scaled else 1.0
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3))
    out, lse = mfa.flash_attn_fwd(q.cuda().to(dtype), k.cuda().to(dtype), v.cuda().to(dtype),
                                  scale=scale, return_lse=True, causal=True)
    qr, kr, vr = (_rounded(x, dtype) for x in (q, k, v))
    assert _maxdiff(out.float().cpu().numpy(), ref_cpu.attention_bhnd(qr, kr, vr, scale=scale, causal=True)) <= LP_ATOL
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(qr, kr, scale=scale, causal=True)) <= 1e-3


# single head with many key shares, several query groups per workgroup, ragged sizes, odd head dims
@pytest.mark.parametrize("shape", [(1, 1, 64, 64), (2, 3, 130, 64), (1, 2, 77, 128), (2, 1, 300, 40), (1, 2, 33, 5),
                                   (1, 1, 1024, 64), (4, 16, 512, 64), (1, 64, 256, 64), (2, 8, 200, 192)])
def test_causal_f32_vs_oracle(mfa, shape):
    B, H, N, d = shape
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3))
    out, lse = mfa.flash_attn_fwd(q.cuda(), k.cuda(), v.cuda(), scale=0.5, return_lse=True, causal=True)
    assert _maxdiff(out.cpu().numpy(), ref_cpu.attention_bhnd(q.numpy(), k.numpy(), v.numpy(), scale=0.5, causal=True)) <= F32_ATOL
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(q.numpy(), k.numpy(), scale=0.5, causal=True)) <= 1e-5


@pytest.mark.parametrize("dtype,d,tol", [(torch.float32, 64, F32_ATOL), (torch.bfloat16, 128, LP_ATOL)])
def test_causal_structure(mfa, dtype, d, tol):
    """What the mask means, independent of any oracle: row 0 is V[0]; nothing after position p
    influences rows <= p; and the first p rows equal the causal result of the length-p prefix."""
    B, H, N, p = 1, 2, 700, 300
    torch.manual_seed(2)
    q, k, v = (torch.randn(B, H, N, d, device="cuda").to(dtype) for _ in range(3))
    out = mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5, causal=True)
    assert float((out[:, :, 0].float() - v[:, :, 0].float()).abs().max()) <= (2 ** -8 if dtype != torch.float32 else 1e-6)
    k2, v2 = k.clone(), v.clone()
    k2[:, :, p + 1:] = torch.randn_like(k2[:, :, p + 1:]) * 3
    v2[:, :, p + 1:] = torch.randn_like(v2[:, :, p + 1:]) * 3
    out2 = mfa.flash_attn_fwd(q, k2, v2, scale=d ** -0.5, causal=True)
    if dtype == torch.float32:
        assert torch.equal(out2[:, :, :p + 1], out[:, :, :p + 1])
    else:
        # not bit-for-bit in low precision: the 32 rows of a wave move their reference max
        # together (deferred rescale), so a later key that is masked for row i can still change
        # WHEN row i's max is refreshed -- the same sum in a different rounding
        assert float((out2[:, :, :p + 1].float() - out[:, :, :p + 1].float()).abs().max()) <= tol
    out_p = mfa.flash_attn_fwd(q[:, :, :p].contiguous(), k[:, :, :p].contiguous(), v[:, :, :p].contiguous(),
                               scale=d ** -0.5, causal=True)
    assert float((out_p.float() - out[:, :, :p].float()).abs().max()) <= tol


def test_causal_headline_size_sampled(mfa, c_oracle):
    """Headline shape under the mask: sampled rows (block edges, diagonal-tile edges, random)
    against the C oracle with each row's own key limit."""
    B, H, N, d = 4, 32, 4096, 128
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    out = mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5, causal=True)
    assert bool(torch.isfinite(out.float()).all())
    gen = np.random.default_rng(1)
    for (b, h) in [(0, 0), (2, 9), (3, 31)]:
        rows = np.unique(np.concatenate([[0, 1, 31, 32, 63, 64, 255, 256, 257, 2047, 2048, N - 1], gen.integers(0, N, 64)]))
        rt = torch.from_numpy(rows).cuda()
        ref = c_oracle.attention(q[b, h, rt].float().cpu().numpy(), k[b, h].float().cpu().numpy(),
                                 v[b, h].float().cpu().numpy(), scale=d ** -0.5, causal_pos=rows)
        assert _maxdiff(out[b, h, rt].float().cpu().numpy(), ref) <= LP_ATOL


def test_causal_flag_errors(mfa):
    import ctypes
    from my_flash_attn_cuda import _lib
    q = torch.randn(1, 1, 64, 64, device="cuda", dtype=torch.bfloat16)
    o = torch.empty_like(q)
    rc = _lib.lib.myfa_attn_fwd_bhnd_ex(q.data_ptr(), q.data_ptr(), q.data_ptr(), o.data_ptr(), None,
                                        1, 1, 64, 64, ctypes.c_float(1.0), 2, 6, None)
    assert rc != 0 and b"flag" in _lib.lib.myfa_last_error()
