"""GPU parity tests for the attention backward (C ABI myfa_attn_bwd_bhnd, Python
flash_attn_bwd / flash_attn_func).

The reference has no backward (its fused forward is a stub,
csrc/my_flash_attn/flash_attention.cu:23-69).  The oracle `ref_cpu.attention_bwd_bhnd` is
pinned by oracle/make_golden.py against autograd of PyTorch's scaled_dot_product_attention in
float64 (to 1e-13), and the fixtures tests/golden/attn_bwd_*.npz hold those autograd results.

Tolerance.  Gradients are not O(1) like the forward's outputs, so the first bound is relative to
each gradient tensor's own largest entry: max|g - oracle| <= 2e-2 * max|oracle| for bfloat16
and float16 alike (the forward's 2e-2, which for its O(1) outputs is the same statement),
with the oracle evaluated on the rounded inputs.  That bound alone would let small entries be
wrong by many times their size, so there is a second, element-wise one: with
e(g) = max over elements of |g - oracle| / (|oracle| + rms(oracle)) -- relative error for the
large entries, error against the tensor's typical magnitude for the small ones -- the kernel's
e must not exceed twice the e of PyTorch's own 16-bit scaled_dot_product_attention backward on
the same inputs (floored at 2e-2, the forward's bar).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

REL = 2e-2


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


def _rel(g, ref):
    """max|g - ref| relative to the largest entry of ref (floored at 1e-2: a gradient that is
    exactly zero, e.g. dq and dk for a single key, is compared absolutely)."""
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(g, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-2))


def _elem(g, ref):
    """max over elements of |g - ref| / (|ref| + rms(ref))"""
    ref = np.asarray(ref, np.float64)
    rms = max(float(np.sqrt(np.mean(ref * ref))), 1e-3)
    return float((np.abs(np.asarray(g, np.float64) - ref) / (np.abs(ref) + rms)).max())


def _torch_lowp_grads(q, k, v, do, dtype, scale, causal):
    """PyTorch's own backward of scaled_dot_product_attention in `dtype` on the GPU."""
    qd, kd, vd = (x.cuda().to(dtype).requires_grad_(True) for x in (q, k, v))
    o = torch.nn.functional.scaled_dot_product_attention(qd, kd, vd, scale=scale, is_causal=causal)
    return torch.autograd.grad(o, (qd, kd, vd), do.cuda().to(dtype))


def _check_elementwise(got, ref, q, k, v, do, dtype, scale, causal):
    theirs = _torch_lowp_grads(q, k, v, do, dtype, scale, causal)
    for nm, a, t, r in zip(("dq", "dk", "dv"), got, theirs, ref):
        ours, torchs = _elem(a.float().cpu().numpy(), r), _elem(t.float().cpu().numpy(), r)
        assert ours <= max(2.0 * torchs, 2e-2), (nm, ours, torchs)


def _rounded(x, dtype):
    return x.to(dtype).to(torch.float32).numpy()


def _run(mfa, q, k, v, do, dtype, scale, causal):
    qd, kd, vd, dod = (x.cuda().to(dtype) for x in (q, k, v, do))
    out, lse = mfa.flash_attn_fwd(qd, kd, vd, scale=scale, return_lse=True, causal=causal)
    return mfa.flash_attn_bwd(dod, qd, kd, vd, out, lse, scale=scale, causal=causal)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["attn_bwd_1x2x100x64", "attn_bwd_1x2x129x128_causal", "attn_bwd_2x1x64x64_causal"])
def test_bwd_vs_autograd_golden(mfa, name, dtype):
    g = load_golden(name)
    q, k, v, do = (torch.from_numpy(g[x]) for x in ("q", "k", "v", "dout"))
    scale, causal = float(g["scale"]), bool(g["causal"])
    got = _run(mfa, q, k, v, do, dtype, scale, causal)
    ref = ref_cpu.attention_bwd_bhnd(*(_rounded(x, dtype) for x in (q, k, v, do)), scale=scale, causal=causal)
    for nm, a, b in zip(("dq", "dk", "dv"), got, ref):
        assert _rel(a.float().cpu().numpy(), b) <= REL, nm
    # and against the stored autograd result of the unrounded inputs, where the rounding of the
    # inputs adds to the distance
    for nm, a in zip(("dq", "dk", "dv"), got):
        assert _rel(a.float().cpu().numpy(), g[nm]) <= 4 * REL, nm


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [
    (1, 1, 1, 64), (1, 2, 63, 128), (2, 3, 200, 64), (1, 2, 256, 128), (1, 2, 257, 128),
    (1, 3, 1025, 64), (2, 2, 1000, 128), (1, 1, 2048, 128),
])
def test_bwd_vs_oracle(mfa, shape, dtype, causal):
    B, H, N, d = shape
    scale = d ** -0.5
    torch.manual_seed(1)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(4))
    got = _run(mfa, q, k, v, do, dtype, scale, causal)
    ref = ref_cpu.attention_bwd_bhnd(*(_rounded(x, dtype) for x in (q, k, v, do)), scale=scale, causal=causal)
    for nm, a, b in zip(("dq", "dk", "dv"), got, ref):
        assert a.shape == shape and a.dtype == dtype
        assert bool(torch.isfinite(a.float()).all()), nm
        assert _rel(a.float().cpu().numpy(), b) <= REL, nm
    if N > 1:
        _check_elementwise(got, ref, q, k, v, do, dtype, scale, causal)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [
    (1, 1, 1, 256), (1, 2, 31, 256), (1, 2, 200, 256), (2, 2, 513, 256), (1, 1, 1024, 256), (1, 3, 129, 192),
    (1, 1, 640, 192), (1, 2, 300, 160), (1, 1, 257, 136), (1, 1, 96, 248),
])
def test_bwd_wide_heads_vs_oracle(mfa, shape, dtype, causal):
    """Head dims above 128: the four-wave build for 192 and 256 columns (others zero-padded up to
    those), ragged lengths on both sides of its 128-row blocks and 32-row tiles."""
    B, H, N, d = shape
    scale = d ** -0.5
    torch.manual_seed(5)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(4))
    got = _run(mfa, q, k, v, do, dtype, scale, causal)
    ref = ref_cpu.attention_bwd_bhnd(*(_rounded(x, dtype) for x in (q, k, v, do)), scale=scale, causal=causal)
    for nm, a, b in zip(("dq", "dk", "dv"), got, ref):
        assert a.shape == shape and a.dtype == dtype
        assert bool(torch.isfinite(a.float()).all()), nm
        assert _rel(a.float().cpu().numpy(), b) <= REL, nm
    if N > 1:
        _check_elementwise(got, ref, q, k, v, do, dtype, scale, causal)


def test_bwd_wide_heads_unscaled_logits(mfa):
    """scale = 1.0 at d = 256: logits with a standard deviation of 16."""
    B, H, N, d = 1, 2, 384, 256
    torch.manual_seed(6)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(4))
    got = _run(mfa, q, k, v, do, torch.bfloat16, 1.0, True)
    ref = ref_cpu.attention_bwd_bhnd(*(_rounded(x, torch.bfloat16) for x in (q, k, v, do)), scale=1.0, causal=True)
    for nm, a, b in zip(("dq", "dk", "dv"), got, ref):
        assert _rel(a.float().cpu().numpy(), b) <= REL, nm
    _check_elementwise(got, ref, q, k, v, do, torch.bfloat16, 1.0, True)


@pytest.mark.parametrize("d", [192, 256])
def test_autograd_function_wide_heads(mfa, d):
    """flash_attn_func is differentiable for head dims 192 and 256: against autograd of PyTorch's
    SDPA in float64 on the same (rounded) inputs."""
    B, H, N = 1, 3, 320
    torch.manual_seed(7)
    q, k, v = (torch.randn(B, H, N, d, device="cuda", dtype=torch.float16, requires_grad=True) for _ in range(3))
    do = torch.randn(B, H, N, d, device="cuda", dtype=torch.float16)
    out = mfa.flash_attn_func(q, k, v, scale=d ** -0.5, causal=False)
    out.backward(do)
    qf, kf, vf = (x.detach().double().cpu().requires_grad_(True) for x in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, scale=d ** -0.5)
    ref.backward(do.double().cpu())
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) <= 2e-2
    for a, b in ((q.grad, qf.grad), (k.grad, kf.grad), (v.grad, vf.grad)):
        assert _rel(a.float().cpu().numpy(), b.numpy()) <= REL


def test_bwd_into_wide_head(mfa):
    """The allocation-free entry serves the same head dims (same kernels: equal bit for bit)."""
    B, H, N, d = 1, 2, 200, 256
    torch.manual_seed(8)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.bfloat16, device="cuda") for _ in range(4))
    out, lse = mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5, return_lse=True)
    want = mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=d ** -0.5)
    got = mfa.flash_attn_bwd_into(do, q, k, v, out, lse, *(torch.empty_like(q) for _ in range(3)), scale=d ** -0.5)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape,causal", [((1, 2, 300, 64), False), ((1, 2, 512, 128), False), ((2, 1, 700, 128), True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bwd_unscaled_logits(mfa, shape, causal, dtype):
    """scale = 1.0 (the reference's semantics for the forward): sharp softmax, large dS; at d = 128
    the logits have a standard deviation of 11."""
    B, H, N, d = shape
    torch.manual_seed(3)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(4))
    got = _run(mfa, q, k, v, do, dtype, 1.0, causal)
    ref = ref_cpu.attention_bwd_bhnd(*(_rounded(x, dtype) for x in (q, k, v, do)), scale=1.0, causal=causal)
    for nm, a, b in zip(("dq", "dk", "dv"), got, ref):
        assert _rel(a.float().cpu().numpy(), b) <= REL, nm
    _check_elementwise(got, ref, q, k, v, do, dtype, 1.0, causal)


def test_autograd_function_matches_torch(mfa):
    """flash_attn_func under torch.autograd against PyTorch's own SDPA backward on the GPU in
    float32 (not the oracle: an end-to-end check of the autograd plumbing)."""
    B, H, N, d = 2, 4, 384, 128
    torch.manual_seed(4)
    q, k, v = (torch.randn(B, H, N, d, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    do = torch.randn(B, H, N, d, device="cuda", dtype=torch.bfloat16)
    out = mfa.flash_attn_func(q, k, v, scale=d ** -0.5, causal=True)
    out.backward(do)
    qf, kf, vf = (x.detach().float().requires_grad_(True) for x in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, is_causal=True, scale=d ** -0.5)
    ref.backward(do.float())
    for a, b in ((q.grad, qf.grad), (k.grad, kf.grad), (v.grad, vf.grad)):
        assert _rel(a.float().cpu().numpy(), b.cpu().numpy()) <= REL


def test_bwd_headline_shape(mfa):
    """Headline shape (bf16, B=4 H=32 N=4096 d=128): one whole head against the float64 oracle
    (heads are independent), and on the full tensors two size-independent identities:
    every gradient is finite, and sum over keys of dV equals sum over queries of dout
    (each row of P sums to one)."""
    B, H, N, d = 4, 32, 4096, 128
    scale = d ** -0.5
    torch.manual_seed(1)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.bfloat16, device="cuda") for _ in range(4))
    out, lse = mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True)
    dq, dk, dv = mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=scale)
    for t in (dq, dk, dv):
        assert bool(torch.isfinite(t.float()).all())
    lhs, rhs = dv.float().sum(dim=2), do.float().sum(dim=2)
    assert float((lhs - rhs).abs().max()) <= REL * float(rhs.abs().max())
    b, h = 2, 17
    ref = ref_cpu.attention_bwd_bhnd(*(x[b, h].float().cpu().numpy() for x in (q, k, v, do)), scale=scale)
    for nm, a, r in zip(("dq", "dk", "dv"), (dq, dk, dv), ref):
        assert _rel(a[b, h].float().cpu().numpy(), r) <= REL, nm


def test_bwd_causal_headline_shape(mfa):
    """The headline shape under the causal mask: two whole heads against the float64 oracle (heads are
    independent), every gradient finite, and the identity that survives the mask: the sum over
    keys of dV equals the sum over queries of dout (each row of P still sums to one)."""
    B, H, N, d = 4, 32, 4096, 128
    scale = d ** -0.5
    torch.manual_seed(2)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.bfloat16, device="cuda") for _ in range(4))
    out, lse = mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True, causal=True)
    dq, dk, dv = mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=scale, causal=True)
    for t in (dq, dk, dv):
        assert bool(torch.isfinite(t.float()).all())
    lhs, rhs = dv.float().sum(dim=2), do.float().sum(dim=2)
    assert float((lhs - rhs).abs().max()) <= REL * float(rhs.abs().max())
    for b, h in ((0, 0), (3, 30)):
        one = [x[b, h][None, None].float().cpu() for x in (q, k, v, do)]
        ref = ref_cpu.attention_bwd_bhnd(*(x[0, 0].numpy() for x in one), scale=scale, causal=True)
        theirs = _torch_lowp_grads(*one, torch.bfloat16, scale, True)
        for nm, a, t, r in zip(("dq", "dk", "dv"), (dq, dk, dv), theirs, ref):
            assert _rel(a[b, h].float().cpu().numpy(), r) <= REL, nm
            ours, torchs = _elem(a[b, h].float().cpu().numpy(), r), _elem(t[0, 0].float().cpu().numpy(), r)
            assert ours <= max(2.0 * torchs, 2e-2), (nm, ours, torchs)


# ---- float32: float32 matrix instructions, any head dim up to 512, any N
#
# Tolerance: max |g - g_oracle| <= 2e-5 * max(|g_oracle|, 1) against the float64 oracle on the same
# float32 inputs (the kernel recomputes exp(scale * S - lse) in float32: an error of a few 1e-6 in
# a logit of magnitude 30 is a relative error of that size in the weight).

F32_BWD_TOL = 2e-5


def _f32_bwd_check(mfa, shape, scale, causal, seed=0):
    B, H, N, d = shape
    torch.manual_seed(seed + 97 * N + d)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(4))
    got = _run(mfa, q, k, v, do, torch.float32, scale, causal)
    ref = ref_cpu.attention_bwd_bhnd(q.numpy(), k.numpy(), v.numpy(), do.numpy(), scale=scale, causal=causal)
    for nm, a, r in zip(("dq", "dk", "dv"), got, ref):
        assert a.shape == shape and a.dtype == torch.float32
        err = np.abs(a.cpu().numpy().astype(np.float64) - r).max()
        assert err <= F32_BWD_TOL * max(np.abs(r).max(), 1.0), (nm, shape, causal, err, np.abs(r).max())


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", [(2, 2, 128, 64), (1, 3, 256, 128), (1, 2, 200, 64), (2, 1, 77, 128), (1, 1, 130, 32),
                                   (1, 2, 64, 16), (1, 1, 1, 64), (1, 1, 17, 8), (1, 2, 300, 256), (1, 1, 96, 192)])
def test_bwd_f32_vs_oracle(mfa, shape, causal):
    _f32_bwd_check(mfa, shape, shape[-1] ** -0.5, causal)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", [(1, 2, 99, 20), (1, 1, 65, 63), (1, 1, 129, 65), (2, 1, 50, 1), (1, 1, 33, 99),
                                   (1, 1, 80, 130), (1, 1, 40, 250), (1, 1, 2, 2)])
def test_bwd_f32_ragged_head_dims(mfa, shape, causal):
    """Head dims that are no multiple of 4 or 16 (the reference's own test dims among them,
    tests/test_flash_attn.py:155): scalar loads, zero-filled fragments, guarded stores."""
    _f32_bwd_check(mfa, shape, shape[-1] ** -0.5, causal)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", [(1, 9, 1024, 64), (3, 5, 520, 128), (1, 1, 2048, 128), (1, 17, 300, 40)])
def test_bwd_f32_many_blocks_and_heads(mfa, shape, causal):
    """Long sequences (many 64-row blocks per head, many streamed blocks) and head counts that are no
    multiple of 8 (the grid is padded to one: the padding workgroups leave in front of every barrier)."""
    _f32_bwd_check(mfa, shape, shape[-1] ** -0.5, causal)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", [(1, 2, 100, 512), (1, 1, 200, 300), (2, 1, 65, 257), (1, 1, 33, 499)])
def test_bwd_f32_head_dims_above_256(mfa, shape, causal):
    """257..512: the output columns go in groups, each in a walk of its own over the other side."""
    _f32_bwd_check(mfa, shape, shape[-1] ** -0.5, causal)


@pytest.mark.parametrize("shape", [(1, 2, 256, 64), (1, 1, 192, 128)])
def test_bwd_f32_unscaled_logits(mfa, shape):
    """scale = 1.0, the reference's semantics: logits of magnitude ~ 3 sqrt(d)."""
    _f32_bwd_check(mfa, shape, 1.0, False)
    _f32_bwd_check(mfa, shape, 1.0, True)


def test_autograd_function_f32(mfa):
    """flash_attn_func differentiates float32 inputs: against autograd of PyTorch's float64
    attention on the same values."""
    torch.manual_seed(5)
    B, H, N, d = 2, 2, 150, 72
    q, k, v = (torch.randn(B, H, N, d, device="cuda", requires_grad=True) for _ in range(3))
    do = torch.randn(B, H, N, d, device="cuda")
    out = mfa.flash_attn_func(q, k, v, scale=d ** -0.5, causal=True)
    got = torch.autograd.grad(out, (q, k, v), do)
    qd, kd, vd = (x.detach().double().requires_grad_(True) for x in (q, k, v))
    ref_out = torch.nn.functional.scaled_dot_product_attention(qd, kd, vd, is_causal=True, scale=d ** -0.5)
    ref = torch.autograd.grad(ref_out, (qd, kd, vd), do.double())
    for a, r in zip(got, ref):
        assert float((a.double() - r).abs().max()) <= F32_BWD_TOL * max(float(r.abs().max()), 1.0)


def test_f32_more_heads_than_a_grid_holds(mfa):
    """B*H above 65535 (the y extent of a launch grid): the float32 forward and backward go in
    successive launches; checked on heads of both launches against PyTorch's float64 attention."""
    torch.manual_seed(8)
    B, H, N, d = 3, 22000, 20, 8            # 66000 heads
    q, k, v = (torch.randn(B, H, N, d, device="cuda", requires_grad=True) for _ in range(3))
    do = torch.randn(B, H, N, d, device="cuda")
    out = mfa.flash_attn_func(q, k, v, scale=d ** -0.5)
    got = torch.autograd.grad(out, (q, k, v), do)
    qd, kd, vd = (x.detach().double().requires_grad_(True) for x in (q, k, v))
    ref_out = torch.nn.functional.scaled_dot_product_attention(qd, kd, vd, scale=d ** -0.5)
    ref = torch.autograd.grad(ref_out, (qd, kd, vd), do.double())
    assert float((out.detach().double() - ref_out.detach()).abs().max()) <= 1e-5
    for a, r in zip(got, ref):
        assert float((a.double() - r).abs().max()) <= F32_BWD_TOL * max(float(r.abs().max()), 1.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bwd_16bit_head_dims_above_256(mfa, dtype):
    """16-bit tensors beyond 256 columns: forward and backward both run the float32 kernels on float32
    copies (host layer); flash_attn_func differentiates them."""
    torch.manual_seed(11)
    B, H, N, d = 1, 2, 90, 320
    scale = d ** -0.5
    q, k, v, do = (torch.randn(B, H, N, d) for _ in range(4))
    qd, kd, vd = (x.cuda().to(dtype).requires_grad_(True) for x in (q, k, v))
    out = mfa.flash_attn_func(qd, kd, vd, scale=scale, causal=True)
    got = torch.autograd.grad(out, (qd, kd, vd), do.cuda().to(dtype))
    ref = ref_cpu.attention_bwd_bhnd(*(_rounded(x, dtype) for x in (q, k, v, do)), scale=scale, causal=True)
    for a, r in zip(got, ref):
        assert a.dtype == dtype and a.shape == (B, H, N, d)
        assert _rel(a.float().cpu().numpy(), r) <= REL


def test_bwd_into_f32(mfa):
    """The entry point without a workspace argument serves float32 as well, bit for bit."""
    torch.manual_seed(6)
    B, H, N, d = 1, 2, 100, 40
    q, k, v, do = (torch.randn(B, H, N, d, device="cuda") for _ in range(4))
    out, lse = mfa.flash_attn_fwd(q, k, v, scale=0.2, return_lse=True)
    want = mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=0.2)
    dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
    mfa.flash_attn_bwd_into(do, q, k, v, out, lse, dq, dk, dv, scale=0.2)
    for a, b in zip((dq, dk, dv), want):
        assert torch.equal(a, b)


def test_bwd_errors(mfa):
    q = torch.randn(1, 1, 64, 64, device="cuda", dtype=torch.float32)
    lse = torch.zeros(1, 1, 64, device="cuda")
    with pytest.raises(RuntimeError, match="float16, bfloat16 and float32"):
        mfa.flash_attn_bwd(*(q.double() for _ in range(5)), lse)
    qb = q.bfloat16()
    with pytest.raises(RuntimeError, match="lse"):
        mfa.flash_attn_bwd(qb, qb, qb, qb, qb, lse[:, :, :32])
    with pytest.raises(RuntimeError, match="Wrong sizes"):
        mfa.flash_attn_bwd(qb, qb, qb[:, :, :32], qb, qb, lse)
    wide = torch.randn(1, 1, 64, 520, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="head dims up to 512"):
        mfa.flash_attn_bwd(wide, wide, wide, wide, wide, lse)
