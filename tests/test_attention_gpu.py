"""GPU parity tests for the attention-forward path, through the drop-in module and the C ABI.

Mirrors the reference's test_attention_kernel / test_flash_attention_kernel
(tests/test_flash_attn.py:154-186): the same nine ragged (N, d) shapes, seed 1,
randn inputs, my_attention(Q, K^T.contiguous(), V) against
scaled_dot_product_attention(scale=1.0) -- here against the committed golden
outputs of the reference's own Python functions and against the float64 oracle.

Tolerances
  float32:  |out - float64 oracle| <= 1e-5.  Against the reference's float32
            golden outputs the bound is the reference's own 1e-4
            (tests/test_flash_attn.py:168,186): those outputs are themselves
            ~1.2e-5 from the exact result at N ~ 1000 (tests/golden/MANIFEST.json),
            so 1e-5 against them is not attainable by any other summation order.
  f16/bf16: |out - float64 oracle on the rounded inputs| <= 2e-2.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

F32_ATOL = 1e-5
REF_ATOL = 1e-4
LP_ATOL = 2e-2
REF_SHAPES = [(2, 2), (1, 1), (2, 20), (2, 99), (20, 2), (99, 20), (1024, 64), (1023, 63), (1025, 65)]


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


def _maxdiff(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


# ------------------------------------------------------------------ 2-D float32 drop-in

@pytest.mark.parametrize("n,d", REF_SHAPES + [(128, 64)])
def test_my_attention_vs_golden(mfa, n, d):
    g = load_golden(f"attn_{n}x{d}")
    q, k, v = (torch.from_numpy(g[x]).cuda() for x in ("q", "k", "v"))
    out = mfa.my_attention(q, k.transpose(0, 1).contiguous(), v)
    assert out.shape == (n, d) and out.dtype == torch.float32
    o = out.cpu().numpy()
    exact = ref_cpu.attention_naive(g["q"], g["k"], g["v"])
    dist = {"float64 oracle": _maxdiff(o, exact)}
    dist.update({key: _maxdiff(o, g[key]) for key in ("out_ref_naive", "out_ref_flash", "out_sdpa")})
    # the measured distances, for the record (pytest -s / -rA shows them): to the exact result, and to the
    # reference's own float32 outputs, which are themselves 1.1-1.4e-5 from exact at n = 1024
    print(f"my_attention {n}x{d}: " + ", ".join(f"|out - {k}| = {v:.3e}" for k, v in dist.items())
          + f"; |out_ref_naive - float64 oracle| = {_maxdiff(g['out_ref_naive'], exact):.3e}")
    assert dist["float64 oracle"] <= F32_ATOL
    for key in ("out_ref_naive", "out_ref_flash", "out_sdpa"):
        assert dist[key] < REF_ATOL


@pytest.mark.parametrize("n,d", [(1, 7), (17, 1), (65, 33), (130, 64), (300, 100), (64, 128), (200, 130), (50, 257)])
def test_my_attention_more_shapes(mfa, n, d):
    torch.manual_seed(1)
    q, k, v = (torch.randn(n, d, dtype=torch.float32) for _ in range(3))
    out = mfa.my_attention(q.cuda(), k.t().contiguous().cuda(), v.cuda())
    assert _maxdiff(out.cpu().numpy(), ref_cpu.attention_naive(q.numpy(), k.numpy(), v.numpy())) <= F32_ATOL


@pytest.mark.parametrize("n,d", [(1024, 64), (1023, 63), (1025, 65), (2048, 128), (2000, 20), (700, 300)])
def test_my_attention_key_slices_with_and_without_workspace(mfa, n, d):
    """Through the C ABI: with a workspace of the advertised size the keys are split over
    workgroups and merged by the second kernel; with none (or too small a one) the call does what
    myfa_attn_fwd_f32 does.  All must meet the float64 oracle to 1e-5, the reference's own
    shapes included (tests/test_flash_attn.py:155)."""
    from my_flash_attn_cuda import _lib
    lib = _lib.lib
    torch.manual_seed(1)
    q, k, v = (torch.randn(n, d, dtype=torch.float32) for _ in range(3))
    exact = ref_cpu.attention_naive(q.numpy(), k.numpy(), v.numpy())
    qd, ktd, vd = q.cuda(), k.t().contiguous().cuda(), v.cuda()
    st = torch.cuda.current_stream().cuda_stream
    nbytes = lib.myfa_attn_fwd_f32_ws_bytes(n, d)
    assert nbytes > 0                                   # these shapes are meant to split
    ws = torch.full((nbytes // 4,), float("nan"), device="cuda")       # scratch may hold anything
    outs = []
    for wptr, wlen in ((ws.data_ptr(), nbytes), (None, 0), (ws.data_ptr(), 64)):
        out = torch.full((n, d), float("nan"), device="cuda")
        _lib.check(lib.myfa_attn_fwd_f32_ws(qd.data_ptr(), ktd.data_ptr(), vd.data_ptr(), out.data_ptr(),
                                            n, d, 1.0, wptr, wlen, st))
        outs.append(out.cpu().numpy())
        assert _maxdiff(outs[-1], exact) <= F32_ATOL
    assert np.array_equal(outs[1], outs[2])             # both fell back to the one-kernel path
    assert lib.myfa_attn_fwd_f32_ws_bytes(16, 64) == 0  # nothing to split


def test_my_attention_errors(mfa):
    q = torch.randn(8, 4, device="cuda")
    with pytest.raises(RuntimeError, match="Wrong sizes"):
        mfa.my_attention(q, q, q)                      # K not transposed
    with pytest.raises(RuntimeError, match="contiguous"):
        mfa.my_attention(q, q.t(), q)                  # transposed view, not contiguous
    with pytest.raises(RuntimeError, match="Float"):
        mfa.my_attention(q.half(), q.t().contiguous().half(), q.half())


def test_matmul_trio(mfa):
    # reference shapes, tests/test_flash_attn.py:106-130
    for (h, k, w) in [(2, 2, 20), (20, 2, 2), (3, 7, 11), (1024, 1024, 1024), (1000, 10, 10), (10, 10, 1000),
                      (999, 999, 999), (1, 1, 1), (65, 17, 129), (64, 16, 64), (300, 1, 5)]:
        torch.manual_seed(1)
        a, b = torch.randn(h, k, device="cuda"), torch.randn(k, w, device="cuda")
        ref = (a.double() @ b.double()).float()
        assert torch.isclose(mfa.my_matmul(a, b), ref, atol=1e-4).all()
        assert torch.isclose(mfa.my_matmul_cublas(a, b), ref, atol=1e-4).all()
        out = torch.empty(h, w, device="cuda")
        assert mfa.my_matmul_out(out, a, b) is out
        assert torch.isclose(out, ref, atol=1e-4).all()
    with pytest.raises(RuntimeError, match="Size mismatch"):
        mfa.my_matmul(torch.randn(3, 4, device="cuda"), torch.randn(5, 6, device="cuda"))


@pytest.mark.parametrize("shape", [(3000, 1000, 2900), (2999, 1001, 3001), (4096, 512, 4096), (8192, 64, 8192)])
def test_matmul_large_blocks(mfa, shape):
    """Shapes that the launcher gives to the 128 x 128 kernel (>= 512 blocks), aligned and not;
    sampled rows against float64, everything against the vendor GEMM."""
    h, k, w = shape
    torch.manual_seed(3)
    a, b = torch.randn(h, k, device="cuda"), torch.randn(k, w, device="cuda")
    out = mfa.my_matmul(a, b)
    rows = torch.tensor([0, 1, 63, 64, 127, 128, h // 2, h - 1], device="cuda")
    ref = (a[rows].double() @ b.double()).float()
    assert torch.isclose(out[rows], ref, atol=2e-4, rtol=1e-5).all()
    assert float((out - torch.matmul(a, b)).abs().max()) <= 1e-3


# ------------------------------------------------------------------ batched entry point

def _rounded(x, dtype):
    return x.to(dtype).float().numpy()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [
    (1, 1, 64, 64), (1, 1, 1, 64), (2, 3, 200, 64), (1, 3, 1025, 64), (1, 2, 256, 128),
    (1, 2, 257, 128), (1, 1, 63, 128), (2, 2, 1000, 128), (1, 5, 511, 128), (3, 1, 320, 64),
])
@pytest.mark.parametrize("scaled", [False, True])
def test_bhnd_lowp_vs_oracle(mfa, dtype, shape, scaled):
    B, H, N, d = shape
    scale = d ** -0.5 if scaled else 1.0
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3))
    out, lse = mfa.flash_attn_fwd(q.cuda().to(dtype), k.cuda().to(dtype), v.cuda().to(dtype),
                                  scale=scale, return_lse=True)
    assert out.shape == shape and out.dtype == dtype
    qr, kr, vr = (_rounded(x, dtype) for x in (q, k, v))
    ref = ref_cpu.attention_bhnd(qr, kr, vr, scale=scale)
    assert _maxdiff(out.float().cpu().numpy(), ref) <= LP_ATOL
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(qr, kr, scale=scale)) <= 1e-3


# the last three make the launcher pick several query groups per workgroup (with one and
# with two key shares) and the 256-column build
@pytest.mark.parametrize("shape", [(1, 1, 64, 64), (2, 3, 130, 64), (1, 2, 77, 128), (2, 1, 300, 40), (1, 2, 33, 5),
                                   (4, 16, 512, 64), (1, 64, 256, 64), (2, 8, 200, 192)])
def test_bhnd_f32_vs_oracle(mfa, shape):
    B, H, N, d = shape
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3))
    out, lse = mfa.flash_attn_fwd(q.cuda(), k.cuda(), v.cuda(), scale=0.5, return_lse=True)
    ref = ref_cpu.attention_bhnd(q.numpy(), k.numpy(), v.numpy(), scale=0.5)
    assert _maxdiff(out.cpu().numpy(), ref) <= F32_ATOL
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(q.numpy(), k.numpy(), scale=0.5)) <= 1e-5


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 128), (torch.float16, 64)])
@pytest.mark.parametrize("N", [1, 100, 191])
def test_memory_behind_the_tensors_never_reaches_the_output(mfa, dtype, d, N):
    """K/V tiles are fetched 64 rows at a time and up to three tiles ahead, so for the last
    head the fetch addresses run past the end of the tensor.  Those fetches must come back as
    zeros (buffer bounds), never as whatever lies behind: Q, K and V sit here in front of a
    NaN-filled tail, and a single 0 * NaN in the P.V product would poison the output."""
    B, H = 1, 2
    torch.manual_seed(1)
    n = B * H * N * d
    tail = 64 * 1024
    bufs = [torch.full((n + tail,), float("nan"), dtype=dtype, device="cuda") for _ in range(3)]
    host = [torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3)]
    views = []
    for b, x in zip(bufs, host):
        b[:n] = x.reshape(-1).to(dtype).cuda()
        views.append(b[:n].view(B, H, N, d))
    out = mfa.flash_attn_fwd(*views, scale=d ** -0.5)
    assert bool(torch.isfinite(out.float()).all())
    qr, kr, vr = (_rounded(x, dtype) for x in host)
    assert _maxdiff(out.float().cpu().numpy(), ref_cpu.attention_bhnd(qr, kr, vr, scale=d ** -0.5)) <= LP_ATOL


def test_bhnd_errors(mfa):
    from my_flash_attn_cuda import _lib
    q = torch.randn(1, 2, 64, 100, device="cuda", dtype=torch.bfloat16)     # not a multiple of 8: the C ABI refuses,
    rc = _lib.lib.myfa_attn_fwd_bhnd(q.data_ptr(), q.data_ptr(), q.data_ptr(), q.data_ptr(), None, 1, 2, 64, 100,
                                     1.0, _lib.MYFA_BF16, None)
    assert rc == 2 and b"head dim" in _lib.lib.myfa_last_error()
    assert mfa.flash_attn_fwd(q, q, q, scale=0.1).shape == q.shape             # the host layer pads
    q = torch.randn(1, 2, 64, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="scale"):
        mfa.flash_attn_fwd(q, q, q, scale=-1.0)
    with pytest.raises(RuntimeError, match="Wrong sizes"):
        mfa.flash_attn_fwd(q, q[:, :1], q)
    with pytest.raises(RuntimeError, match="mismatch"):
        mfa.flash_attn_fwd(q, q.half(), q)


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 128), (torch.float16, 64), (torch.float16, 128), (torch.bfloat16, 64)])
def test_rescale_branch_is_forced(mfa, dtype, d, scaled):
    """Online-softmax rescale: plant keys that raise single rows' max late in the sweep
    (at chosen tiles), so exp(m_old - m_new) << 1 is exercised, not just ~1.  Unscaled logits run
    the build that looks for every tile's max; at d^-0.5 the library runs the build that forms
    the weights against the standing max and goes back when a row sum passes its limit (2^30 in
    bfloat16, 2^12 in float16): the planted logits, 4 |q|^2 d^-0.5 log2(e) = 46 (d = 64) or 65
    (d = 128) base-2 units above the rest, send it back at the chosen tiles."""
    torch.manual_seed(1)
    N = 640
    scale = d ** -0.5 if scaled else 1.0
    q, k, v = (torch.randn(1, 1, N, d) for _ in range(3))
    for row, key in [(0, 130), (5, 639), (31, 64), (32, 65), (100, 300), (255, 575), (256, 1), (639, 638)]:
        k[0, 0, key] = q[0, 0, row] * (4.0 if scaled else 3.0)          # logit ~ 3 |q|^2 >> the rest
    qd, kd, vd = (x.cuda().to(dtype) for x in (q, k, v))
    out, lse = mfa.flash_attn_fwd(qd, kd, vd, scale=scale, return_lse=True)
    qr, kr, vr = _rounded(q, dtype), _rounded(k, dtype), _rounded(v, dtype)
    assert _maxdiff(out.float().cpu().numpy(), ref_cpu.attention_bhnd(qr, kr, vr, scale=scale)) <= LP_ATOL
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(qr, kr, scale=scale)) <= 2e-3


def _sampled_check(mfa, c_oracle, B, H, N, d, dtype, scale, heads, nrows, seed=1):
    torch.manual_seed(seed)
    q, k, v = (torch.randn(B, H, N, d, dtype=dtype, device="cuda") for _ in range(3))
    out = mfa.flash_attn_fwd(q, k, v, scale=scale)
    assert bool(torch.isfinite(out.float()).all())
    gen = np.random.default_rng(seed)
    worst = 0.0
    for (b, h) in heads:
        rows = np.unique(np.concatenate([[0, 31, 32, 255, 256, N - 1], gen.integers(0, N, nrows)]))
        rt = torch.from_numpy(rows).cuda()
        ref = c_oracle.attention(q[b, h, rt].float().cpu().numpy(), k[b, h].float().cpu().numpy(),
                                 v[b, h].float().cpu().numpy(), scale=scale)
        worst = max(worst, _maxdiff(out[b, h, rt].float().cpu().numpy(), ref))
    return q, k, v, out, worst


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 256), (torch.float16, 192), (torch.bfloat16, 136)])
def test_wide_head_dims_sampled(mfa, c_oracle, dtype, d):
    """Head dims above 128 (the 256-column build, one wave per SIMD): B=2 H=8 N=2048, sampled rows
    against all keys, unscaled (saturating) and scaled logits; rows past the last full query block."""
    for scale, n in ((d ** -0.5, 2048), (0.25, 1000)):
        *_, worst = _sampled_check(mfa, c_oracle, 2, 8, n, d, dtype, scale, heads=[(0, 0), (1, 7)], nrows=64)
        assert worst <= LP_ATOL


def test_wide_rescale_branch_is_forced(mfa):
    """The deferred rescale of the 256-column build: keys that raise single rows' max by far more
    than the threshold, late in the sweep."""
    torch.manual_seed(1)
    N, d = 640, 256
    q, k, v = (torch.randn(1, 1, N, d) for _ in range(3))
    for row, key in [(0, 130), (5, 639), (31, 64), (32, 65), (100, 300), (127, 575), (128, 1), (639, 638)]:
        k[0, 0, key] = q[0, 0, row] * 3.0
    qd, kd, vd = (x.cuda().to(torch.bfloat16) for x in (q, k, v))
    out = mfa.flash_attn_fwd(qd, kd, vd, scale=0.5)
    ref = ref_cpu.attention_bhnd(*(_rounded(x, torch.bfloat16) for x in (q, k, v)), scale=0.5)
    assert _maxdiff(out.float().cpu().numpy(), ref) <= LP_ATOL


def test_config3_f16_full_size(mfa, c_oracle):
    """The f16 benchmark configuration: B=8 H=16 N=1024 d=64."""
    for scale in (1.0, 64 ** -0.5):
        *_, worst = _sampled_check(mfa, c_oracle, 8, 16, 1024, 64, torch.float16, scale,
                                   heads=[(0, 0), (3, 7), (7, 15)], nrows=200)
        assert worst <= LP_ATOL


def test_headline_bf16_full_size(mfa, c_oracle):
    """The headline benchmark configuration: bf16, B=4 H=32 N=4096 d=128 -- sampled rows against all 4096 keys,
    plus size-independent properties on the full tensors."""
    B, H, N, d = 4, 32, 4096, 128
    for scale in (1.0, d ** -0.5):
        q, k, v, out, worst = _sampled_check(mfa, c_oracle, B, H, N, d, torch.bfloat16, scale,
                                             heads=[(0, 0), (1, 17), (3, 31)], nrows=96)
        assert worst <= LP_ATOL
    # property 1: a convex combination of V rows stays inside V's range, per head and column
    vmin = v.float().amin(dim=2, keepdim=True) - 2e-2
    vmax = v.float().amax(dim=2, keepdim=True) + 2e-2
    assert bool(((out.float() >= vmin) & (out.float() <= vmax)).all())
    # property 2: permuting the keys (K and V rows together) does not change the result
    perm = torch.randperm(N, device="cuda")
    out_p = mfa.flash_attn_fwd(q, k[:, :, perm].contiguous(), v[:, :, perm].contiguous(), scale=scale)
    assert float((out_p.float() - out.float()).abs().max()) <= LP_ATOL
    # property 3: constant V gives that constant back (weights sum to 1)
    vc = torch.full_like(v, 0.75)
    out_c = mfa.flash_attn_fwd(q, k, vc, scale=scale)
    assert float((out_c.float() - 0.75).abs().max()) <= 2 ** -8
    # property 4: linear in V
    v2 = torch.randn_like(v)
    o1 = mfa.flash_attn_fwd(q, k, v2, scale=scale).float()
    o12 = mfa.flash_attn_fwd(q, k, (v + v2), scale=scale).float()
    assert float((o12 - (out.float() + o1)).abs().max()) <= 3 * LP_ATOL


def test_config5_shard_bf16(mfa, c_oracle):
    """The scaling configuration, the shard one of 8 GPUs gets: bf16, (B*H)/8 = 32 heads, N=16384, d=128."""
    *_, worst = _sampled_check(mfa, c_oracle, 1, 32, 16384, 128, torch.bfloat16, 128 ** -0.5,
                               heads=[(0, 0), (0, 31)], nrows=48)
    assert worst <= LP_ATOL


def test_config5_full_single_gpu(mfa, c_oracle):
    """The scaling configuration whole on one GPU (its 1-GPU point): bf16, B=8 H=32 N=16384 d=128,
    256 heads = 16384 workgroups, 1 GiB per tensor; sampled rows of three heads against all keys."""
    *_, worst = _sampled_check(mfa, c_oracle, 8, 32, 16384, 128, torch.bfloat16, 128 ** -0.5,
                               heads=[(0, 0), (4, 13), (7, 31)], nrows=24)
    assert worst <= LP_ATOL


def test_nondefault_stream_and_reuse(mfa):
    torch.manual_seed(1)
    q, k, v = (torch.randn(2, 4, 512, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    ref = mfa.flash_attn_fwd(q, k, v, scale=0.1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        outs = [mfa.flash_attn_fwd(q, k, v, scale=0.1) for _ in range(3)]
    s.synchronize()
    for o in outs:
        assert torch.equal(o, ref)      # deterministic: same bits every launch, any stream
