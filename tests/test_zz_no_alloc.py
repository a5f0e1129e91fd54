"""GPU test of the allocation-free backward entry point (C ABI myfa_attn_bwd_bhnd_static_ws).

It runs the kernels of myfa_attn_bwd_bhnd with the [2, B, H, N] float workspace in a device
array inside the library, so its gradients must equal that entry point's bit for bit
(deterministic kernels, no atomics).  Both are called through the C ABI with raw pointers;
two calls in a row on one GPU, the second with a smaller input, so that what the first call
left in the shared array is still there when the second runs.  Collected last: a failure here
must not hide the parity tests.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CAUSAL = 1


@pytest.fixture(scope="module")
def mfa():
    import my_flash_attn_cuda
    assert torch.cuda.is_available()
    return my_flash_attn_cuda


def _both(mfa, shape, dtype, scale, causal, seed):
    from my_flash_attn_cuda import _lib
    lib = _lib.lib
    B, H, N, d = shape
    torch.manual_seed(seed)
    q, k, v, do = (torch.randn(B, H, N, d, device="cuda", dtype=dtype) for _ in range(4))
    out, lse = mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True, causal=causal)
    tag = {torch.float16: _lib.MYFA_F16, torch.bfloat16: _lib.MYFA_BF16}[dtype]
    flags = CAUSAL if causal else 0
    stream = torch.cuda.current_stream().cuda_stream
    # existing entry point: caller workspace
    ws = torch.empty(2 * B * H * N, device="cuda", dtype=torch.float32)
    ref = [torch.full_like(q, float("nan")) for _ in range(3)]
    _lib.check(lib.myfa_attn_bwd_bhnd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                      do.data_ptr(), ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr(),
                                      ws.data_ptr(), B, H, N, d, scale, tag, flags, stream))
    # new entry point: outputs allocated here, before the call; nothing but pointers goes in
    got = [torch.full_like(q, float("nan")) for _ in range(3)]
    torch.cuda.synchronize()
    _lib.check(lib.myfa_attn_bwd_bhnd_static_ws(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                do.data_ptr(), got[0].data_ptr(), got[1].data_ptr(), got[2].data_ptr(),
                                                B, H, N, d, scale, tag, flags, stream))
    torch.cuda.synchronize()
    return got, ref


def _assert_bit_equal(got, ref):
    for name, a, b in zip(("dq", "dk", "dv"), got, ref):
        a16 = a.view(torch.int16).cpu().numpy()
        b16 = b.view(torch.int16).cpu().numpy()
        assert bool(torch.isfinite(a.float()).all()), name
        assert np.array_equal(a16, b16), name


def test_static_workspace_two_calls_second_smaller(mfa):
    # first call fills a large part of the array, the second (smaller, other dtype and mask)
    # runs over its leftovers
    got, ref = _both(mfa, (2, 4, 1000, 128), torch.bfloat16, 128 ** -0.5, False, seed=11)
    _assert_bit_equal(got, ref)
    got, ref = _both(mfa, (1, 3, 257, 64), torch.float16, 64 ** -0.5, True, seed=12)
    _assert_bit_equal(got, ref)


def test_static_workspace_python_wrapper_and_oracle(mfa):
    """flash_attn_bwd_into (the host-layer face of the entry point) against the float64 oracle,
    to the backward's own bar."""
    from oracle import ref_cpu
    B, H, N, d = 1, 2, 200, 64
    torch.manual_seed(5)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(4))
    qd, kd, vd, dod = (x.cuda().to(torch.bfloat16) for x in (q, k, v, do))
    out, lse = mfa.flash_attn_fwd(qd, kd, vd, scale=d ** -0.5, return_lse=True)
    dq, dk, dv = (torch.empty_like(qd) for _ in range(3))
    mfa.flash_attn_bwd_into(dod, qd, kd, vd, out, lse, dq, dk, dv, scale=d ** -0.5)
    ref = ref_cpu.attention_bwd_bhnd(*(x.to(torch.bfloat16).float().numpy() for x in (q, k, v, do)), scale=d ** -0.5)
    for nm, a, b in zip(("dq", "dk", "dv"), (dq, dk, dv), ref):
        err = np.abs(a.float().cpu().numpy() - b).max() / max(np.abs(b).max(), 1e-2)
        assert err <= 2e-2, nm


def test_static_workspace_size_limit(mfa):
    from my_flash_attn_cuda import _lib
    lib = _lib.lib
    x = torch.zeros(16, device="cuda", dtype=torch.bfloat16)
    p = x.data_ptr()
    # B*H*N = 2^20 + 2^14 rows: rejected before anything is launched (the pointers are never used)
    rc = lib.myfa_attn_bwd_bhnd_static_ws(p, p, p, p, p, p, p, p, p, 1, 65, 16384, 128, 1.0, _lib.MYFA_BF16, 0, None)
    assert rc == 2
    assert b"static workspace" in lib.myfa_last_error()
    rc = lib.myfa_attn_bwd_bhnd_static_ws(p, p, p, p, p, p, p, p, p, 0, 65, 16384, 128, 1.0, _lib.MYFA_BF16, 0, None)
    assert rc == 0   # empty problem: no-op
