"""GPU parity tests of the kernel variants the launchers choose between.

The shipped library takes no knobs; its tuning flavour (libmyfa_gfx950_tuning.so, the same
sources with -DMYFA_TUNING) can force a variant.  Every variant must meet the oracle on its
own: the two MFMA shapes of the f16/bf16 attention forward (32x32x16 and 16x16x32), the
one-barrier schedule, the deferred-max threshold (0 = the textbook recurrence of the
reference's Python statement, tests/test_flash_attn.py:82-93), and both block sizes of the
float32 matmul (the reference's my_matmul, csrc/my_flash_attn/matmul.cu:22-50).
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import ref_cpu

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

LP_ATOL = 2e-2


@pytest.fixture(scope="module")
def tuned():
    from tuned import Tuned
    assert torch.cuda.is_available()
    t = Tuned()
    yield t
    t.set()


def _rounded(x, dtype):
    return x.to(dtype).float().numpy()


def _maxdiff(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


# (knobs, scale: None = d^-0.5, bound on |O - oracle|)
VARIANTS = [
    (dict(attn_variant=1, attn_thr=0), 1.0, LP_ATOL), (dict(attn_variant=1, attn_thr=8), 1.0, LP_ATOL),
    (dict(attn_variant=0), 1.0, LP_ATOL),
    (dict(attn_variant=2, attn_spec=0, attn_thr=0), 1.0, LP_ATOL), (dict(attn_variant=2, attn_spec=0, attn_thr=8), 1.0, LP_ATOL),
    # what the library picks by itself: the vote at thr = 8 for unscaled logits, the speculative build
    # (weights against the standing max, the max moved when a row sum passes 2^thr) at d^-0.5
    (dict(attn_variant=2), 1.0, LP_ATOL), (dict(attn_variant=2), None, LP_ATOL),
    # the speculative build forced: limit 1 (it goes back in nearly every tile: the textbook recurrence),
    # 2^5, and its own limits; at d^-0.5 the planted keys still jump by ~49 base-2 units, past any limit
    (dict(attn_variant=2, attn_spec=1, attn_thr=0), 1.0, LP_ATOL), (dict(attn_variant=2, attn_spec=1, attn_thr=5), None, LP_ATOL),
    (dict(attn_variant=2, attn_spec=1), None, LP_ATOL), (dict(attn_variant=2, attn_spec=1, attn_thr=60), None, LP_ATOL),
    # ... and on unscaled logits, where the library itself does not use it: one key dominates a row, its
    # weight is 2^x against a stale max instead of the exact 1, and the rounding of that weight to the
    # tensor type (relative 2^-8 in bfloat16, times |v| up to ~5) adds to the output's own rounding
    (dict(attn_variant=2, attn_spec=1), 1.0, 4e-2), (dict(attn_variant=2, attn_spec=1, attn_thr=60), 1.0, 4e-2),
    # the resident-workgroup build (a workgroup streams through a list of query blocks): the library's
    # choices, both softmax forms forced, and two workgroups for the 16 query blocks of this problem
    (dict(attn_variant=3), 1.0, LP_ATOL), (dict(attn_variant=3), None, LP_ATOL),
    (dict(attn_variant=3, attn_spec=0, attn_thr=0), 1.0, LP_ATOL), (dict(attn_variant=3, attn_spec=1, attn_pre=0, attn_thr=0), 1.0, LP_ATOL),
    (dict(attn_variant=3, attn_spec=1, attn_thr=5, attn_wgs=2), None, LP_ATOL), (dict(attn_variant=3, attn_spec=0, attn_wgs=3), 1.0, LP_ATOL),
    # ... with and without Q pre-multiplied by scale * log2(e) and the running max folded into the
    # accumulators' start value (the library's choice wherever it picks the speculative softmax)
    (dict(attn_variant=3, attn_spec=1, attn_pre=0), None, LP_ATOL), (dict(attn_variant=3, attn_spec=1, attn_pre=1), None, LP_ATOL),
    (dict(attn_variant=3, attn_spec=1, attn_pre=1, attn_thr=0, attn_wgs=3), None, LP_ATOL),
    (dict(attn_variant=3, attn_spec=1, attn_pre=1, attn_thr=60, attn_wgs=5), None, LP_ATOL),
    # ... and the speculative softmax whose row sums come out of the matrix cores (sums of the ROUNDED
    # weights, so a dominant key's rounding cancels in O: the library's choice for unscaled bfloat16
    # logits at head dims above 64), at its own limit, at limits that send it back often, and on scaled logits
    (dict(attn_variant=3, attn_spec=2), 1.0, LP_ATOL), (dict(attn_variant=3, attn_spec=2, attn_thr=60), 1.0, LP_ATOL),
    (dict(attn_variant=3, attn_spec=2, attn_thr=0, attn_wgs=3), 1.0, LP_ATOL), (dict(attn_variant=3, attn_spec=2, attn_thr=5), None, LP_ATOL),
    (dict(attn_variant=3, attn_spec=2), None, LP_ATOL),
]


@pytest.mark.parametrize("knobs,scale,atol", VARIANTS,
                         ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()) if isinstance(k, dict) else str(k))
@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 128), (torch.float16, 64), (torch.float16, 128), (torch.bfloat16, 64)])
def test_schedules_shapes_and_deferred_max_agree(tuned, knobs, scale, atol, dtype, d):
    """Random data plus keys that force large, late jumps of single rows' max."""
    tuned.set(**knobs)
    torch.manual_seed(1)
    N = 1000
    scale = d ** -0.5 if scale is None else scale
    q, k, v = (torch.randn(2, 2, N, d) for _ in range(3))
    for row, key in [(0, 130), (5, 999), (15, 3), (16, 700), (31, 64), (32, 65), (100, 300), (255, 575), (256, 1), (999, 998)]:
        k[:, :, key] = q[:, :, row] * 3.0
    qd, kd, vd = (x.cuda().to(dtype) for x in (q, k, v))
    out, lse = tuned.flash_attn_fwd(qd, kd, vd, scale=scale, return_lse=True)
    qr, kr, vr = (_rounded(x, dtype) for x in (q, k, v))
    ref = ref_cpu.attention_bhnd(qr, kr, vr, scale=scale)
    assert _maxdiff(out.float().cpu().numpy(), ref) <= atol
    # with Q pre-multiplied by scale * log2(e) (forced here: the library does not do it when the row
    # statistic is asked for) a logit of 50 moves by up to ~2^-8 of its size, and the statistic with it
    # (and with the row sums taken from the rounded weights, forced here likewise, by 2^-9)
    lse_tol = 6e-2 if knobs.get("attn_pre") == 1 else 4e-3 if knobs.get("attn_spec") == 2 else 2e-3
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(qr, kr, scale=scale)) <= lse_tol
    # the call without the row statistic (where the library may take a faster route) meets the same bound
    assert _maxdiff(tuned.flash_attn_fwd(qd, kd, vd, scale=scale).float().cpu().numpy(), ref) <= atol


@pytest.mark.parametrize("dtype,thr", [(torch.float16, -1), (torch.float16, 15), (torch.bfloat16, -1), (torch.bfloat16, 12)])
def test_matrix_core_row_sums_go_back_in_time(tuned, dtype, thr):
    """Unscaled logits at N = 4096: some row's max jumps by more than the limit in a late tile, so the
    go-back test of the speculative softmax with matrix-core row sums (attn_spec = 2) decides whether
    a float16 weight overflows.  Finite, inside the bound on sampled rows, and the same bits on a
    second run (the test once read exp2 results one issue slot early: non-finite rows that changed
    from run to run)."""
    B, H, N, d = 1, 4, 4096, 128
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=dtype, device="cuda") for _ in range(3))
    tuned.set(attn_variant=3, attn_spec=2, attn_thr=thr)
    out = tuned.flash_attn_fwd(q, k, v, scale=1.0)
    again = tuned.flash_attn_fwd(q, k, v, scale=1.0)
    assert bool(torch.isfinite(out.float()).all())
    assert torch.equal(out, again)
    rows = torch.arange(0, N, 37)
    for h in range(H):
        qr, kr, vr = q[0, h, rows].float().cpu().numpy(), k[0, h].float().cpu().numpy(), v[0, h].float().cpu().numpy()
        # float64 softmax(q k^T) v of the sampled rows against all keys (ref_cpu.attention_naive's math on a
        # rectangular problem)
        sc = qr.astype(np.float64) @ kr.astype(np.float64).T
        w = np.exp(sc - sc.max(axis=1, keepdims=True))
        ref = (w / w.sum(axis=1, keepdims=True)) @ vr.astype(np.float64)
        assert _maxdiff(out[0, h, rows].float().cpu().numpy(), ref) <= LP_ATOL


@pytest.mark.parametrize("variant,spec", [(1, -1), (2, 0), (2, 1), (3, 0), (3, 1), (3, 2), (3, -1)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [
    (1, 1, 64, 64), (1, 1, 1, 64), (2, 3, 200, 64), (1, 3, 1025, 64), (1, 2, 256, 128),
    (1, 2, 257, 128), (1, 1, 63, 128), (2, 2, 1000, 128), (1, 5, 511, 128), (3, 1, 320, 64),
])
def test_mfma_shapes_vs_oracle(tuned, variant, spec, causal, dtype, shape):
    B, H, N, d = shape
    scale = d ** -0.5
    tuned.set(attn_variant=variant, attn_spec=spec)
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3))
    out, lse = tuned.flash_attn_fwd(q.cuda().to(dtype), k.cuda().to(dtype), v.cuda().to(dtype),
                                    scale=scale, causal=causal, return_lse=True)
    qr, kr, vr = (_rounded(x, dtype) for x in (q, k, v))
    ref = ref_cpu.attention_bhnd(qr, kr, vr, scale=scale, causal=causal)
    assert bool(torch.isfinite(out.float()).all())
    assert _maxdiff(out.float().cpu().numpy(), ref) <= LP_ATOL
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(qr, kr, scale=scale, causal=causal)) <= (3e-3 if spec == 2 else 1e-3)


@pytest.mark.parametrize("wgs,occ", [(1, 0), (3, 0), (8, 2), (11, 0), (0, 0), (0, 2)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape,dtype", [
    ((2, 3, 64, 64), torch.float16), ((1, 5, 40, 128), torch.bfloat16), ((2, 2, 700, 128), torch.bfloat16),
    ((1, 3, 1300, 64), torch.float16), ((3, 2, 257, 128), torch.float16), ((1, 4, 130, 64), torch.bfloat16),
    ((2, 2, 1, 128), torch.bfloat16), ((1, 2, 520, 72), torch.bfloat16),
])
def test_resident_workgroups_stream_through_their_items(tuned, wgs, occ, causal, shape, dtype):
    """The resident-workgroup build with FEWER workgroups than query blocks: every workgroup works
    through several (head, query block) items back to back -- rings, Q fragments and the running
    softmax state turn over between items, also for items of a single tile (N <= 64) and under the
    causal mask, where items differ in length.  occ = 2: head dims up to 64 with two workgroups per CU
    (128 registers per lane) instead of the library's one."""
    B, H, N, d = shape
    scale = d ** -0.5
    tuned.set(attn_variant=3, attn_wgs=wgs, attn_occ=occ)
    torch.manual_seed(3)
    q, k, v = (torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3))
    qd, kd, vd = (x.cuda().to(dtype) for x in (q, k, v))
    out, lse = tuned.flash_attn_fwd(qd, kd, vd, scale=scale, causal=causal, return_lse=True)
    qr, kr, vr = (_rounded(x, dtype) for x in (q, k, v))
    ref = ref_cpu.attention_bhnd(qr, kr, vr, scale=scale, causal=causal)
    assert bool(torch.isfinite(out.float()).all())
    assert _maxdiff(out.float().cpu().numpy(), ref) <= LP_ATOL
    assert _maxdiff(lse.cpu().numpy(), ref_cpu.attention_lse(qr, kr, scale=scale, causal=causal)) <= 1e-3
    # without the row statistic the library pre-multiplies Q by scale * log2(e) (unmasked launches)
    out2 = tuned.flash_attn_fwd(qd, kd, vd, scale=scale, causal=causal)
    assert bool(torch.isfinite(out2.float()).all())
    assert _maxdiff(out2.float().cpu().numpy(), ref) <= LP_ATOL


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_memory_behind_the_tensors_never_reaches_the_output(tuned, variant):
    """Fetches past the end of K/V must come back as zeros (buffer bounds), never as the NaNs
    planted behind the tensors."""
    tuned.set(attn_variant=variant)
    B, H, N, d, dtype = 1, 2, 100, 128, torch.bfloat16
    torch.manual_seed(1)
    n = B * H * N * d
    bufs = [torch.full((n + 64 * 1024,), float("nan"), dtype=dtype, device="cuda") for _ in range(3)]
    host = [torch.randn(B, H, N, d, dtype=torch.float32) for _ in range(3)]
    views = []
    for b, x in zip(bufs, host):
        b[:n] = x.reshape(-1).to(dtype).cuda()
        views.append(b[:n].view(B, H, N, d))
    out = tuned.flash_attn_fwd(*views, scale=d ** -0.5)
    assert bool(torch.isfinite(out.float()).all())
    qr, kr, vr = (_rounded(x, dtype) for x in host)
    assert _maxdiff(out.float().cpu().numpy(), ref_cpu.attention_bhnd(qr, kr, vr, scale=d ** -0.5)) <= LP_ATOL


@pytest.mark.parametrize("block", [64, 128])
@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 7, 11), (128, 16, 128), (129, 17, 131), (260, 36, 132), (100, 200, 300),
                                   (256, 4, 512), (130, 1000, 4), (512, 520, 388)])
def test_matmul_both_kernels_small_shapes(tuned, block, shape):
    """Both block sizes forced on ragged and aligned shapes (vector and scalar staging)."""
    tuned.set(matmul_block=block)
    h, k, w = shape
    torch.manual_seed(2)
    a, b = torch.randn(h, k, device="cuda"), torch.randn(k, w, device="cuda")
    ref = (a.double() @ b.double()).float()
    assert torch.isclose(tuned.my_matmul(a, b), ref, atol=1e-4, rtol=1e-5).all()
