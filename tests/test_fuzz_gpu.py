"""Seeded random-shape sweep of every attention entry point against the float64 oracle:
ragged sequence lengths around the tile edges (64-key tiles, 256-row query blocks, 128-key
blocks of the backward, 16-row groups of the float32 kernel), both head dims of the matrix-core
path, odd head dims of the float32 path, masked and unmasked, scaled and unscaled.
Tolerances are the ones stated in the per-path test files.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu

pytestmark = pytest.mark.gpu

EDGE_N = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 300, 383, 511, 512, 513, 700]


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    return my_flash_attn_cuda


def _rounded(x, dtype):
    return x.to(dtype).to(torch.float32).numpy()


def _cases(seed, count, dims):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.choice(EDGE_N)) if rng.random() < 0.7 else int(rng.integers(1, 900))
        out.append((int(rng.integers(1, 3)), int(rng.integers(1, 4)), n, int(rng.choice(dims)),
                    bool(rng.integers(0, 2)), float(rng.choice([1.0, 0.5, 0.125])), int(rng.integers(0, 1 << 30))))
    return out


# 64, 128 and 256 are the native builds; the other multiples of 8 up to 256 run on them with zero
# columns; the reference's own odd head dims (tests/test_flash_attn.py:155: 1, 2, 20, 63, 65, 99)
# and those above 256 go through the host layer's padded / float32 detours
@pytest.mark.parametrize("case", _cases(11, 24, [64, 128]) + _cases(15, 28, [8, 32, 40, 96, 104, 120, 256, 1, 2, 20, 63, 65, 99])
                         + _cases(17, 24, [136, 160, 192, 200, 248, 256, 130, 257, 264]))
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fuzz_forward_lowp(mfa, dtype, case):
    B, H, N, d, causal, scale, seed = case
    if scale == 1.0 and d >= 96:
        scale = 0.25          # unscaled logits at d = 128 saturate the softmax; covered by the fixed-shape tests
    torch.manual_seed(seed)
    q, k, v = (torch.randn(B, H, N, d) for _ in range(3))
    out, lse = mfa.flash_attn_fwd(q.cuda().to(dtype), k.cuda().to(dtype), v.cuda().to(dtype),
                                  scale=scale, return_lse=True, causal=causal)
    qr, kr, vr = (_rounded(x, dtype) for x in (q, k, v))
    assert np.abs(out.float().cpu().numpy() - ref_cpu.attention_bhnd(qr, kr, vr, scale=scale, causal=causal)).max() <= 2e-2
    assert np.abs(lse.cpu().numpy() - ref_cpu.attention_lse(qr, kr, scale=scale, causal=causal)).max() <= 1e-3


@pytest.mark.parametrize("case", _cases(12, 24, [1, 5, 16, 40, 64, 96, 128, 130, 200, 257]))
def test_fuzz_forward_f32(mfa, case):
    B, H, N, d, causal, scale, seed = case
    torch.manual_seed(seed)
    q, k, v = (torch.randn(B, H, N, d) for _ in range(3))
    out, lse = mfa.flash_attn_fwd(q.cuda(), k.cuda(), v.cuda(), scale=scale * 0.5, return_lse=True, causal=causal)
    assert np.abs(out.cpu().numpy() - ref_cpu.attention_bhnd(q.numpy(), k.numpy(), v.numpy(), scale=scale * 0.5, causal=causal)).max() <= 1e-5
    assert np.abs(lse.cpu().numpy() - ref_cpu.attention_lse(q.numpy(), k.numpy(), scale=scale * 0.5, causal=causal)).max() <= 1e-5


# (192 and 256: the four-wave builds; the other dims above 128 are zero-padded up to those)
@pytest.mark.parametrize("case", _cases(13, 16, [64, 128]) + _cases(16, 8, [32, 96, 40]) + _cases(18, 14, [192, 256, 136, 200, 248]))
def test_fuzz_backward(mfa, case):
    B, H, N, d, causal, _, seed = case
    scale = d ** -0.5
    dtype = torch.bfloat16 if seed & 1 else torch.float16
    torch.manual_seed(seed)
    q, k, v, do = (torch.randn(B, H, N, d) for _ in range(4))
    qd, kd, vd, dod = (x.cuda().to(dtype) for x in (q, k, v, do))
    out, lse = mfa.flash_attn_fwd(qd, kd, vd, scale=scale, return_lse=True, causal=causal)
    got = mfa.flash_attn_bwd(dod, qd, kd, vd, out, lse, scale=scale, causal=causal)
    ref = ref_cpu.attention_bwd_bhnd(*(_rounded(x, dtype) for x in (q, k, v, do)), scale=scale, causal=causal)
    for g, r in zip(got, ref):
        assert np.abs(g.float().cpu().numpy() - r).max() <= 2e-2 * max(np.abs(r).max(), 1e-2)


@pytest.mark.parametrize("case", _cases(19, 20, [1, 3, 16, 20, 40, 63, 64, 65, 99, 128, 130, 200, 256]) + _cases(20, 6, [257, 300, 384, 512]))
def test_fuzz_backward_f32(mfa, case):
    """float32 backward (any head dim up to 512): 2e-5 of the largest gradient against the float64 oracle."""
    B, H, N, d, causal, _, seed = case
    scale = d ** -0.5
    torch.manual_seed(seed)
    q, k, v, do = (torch.randn(B, H, N, d) for _ in range(4))
    qd, kd, vd, dod = (x.cuda() for x in (q, k, v, do))
    out, lse = mfa.flash_attn_fwd(qd, kd, vd, scale=scale, return_lse=True, causal=causal)
    got = mfa.flash_attn_bwd(dod, qd, kd, vd, out, lse, scale=scale, causal=causal)
    ref = ref_cpu.attention_bwd_bhnd(q.numpy(), k.numpy(), v.numpy(), do.numpy(), scale=scale, causal=causal)
    for g, r in zip(got, ref):
        assert np.abs(g.cpu().numpy() - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0)


@pytest.mark.parametrize("case", _cases(14, 12, [1, 7, 64, 65, 128, 300]))
def test_fuzz_my_attention_2d(mfa, case):
    _, _, N, d, _, _, seed = case
    torch.manual_seed(seed)
    q, k, v = (torch.randn(N, d) for _ in range(3))
    q, k = q * 0.3, k * 0.3                       # my_attention is unscaled: keep the logits moderate at large d
    ref = ref_cpu.attention_naive(q.numpy(), k.numpy(), v.numpy())
    out = mfa.my_attention(q.cuda(), k.t().contiguous().cuda(), v.cuda())
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-5
