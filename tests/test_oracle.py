"""CPU tests: the oracle against the golden vectors made from the reference.

The fixtures in tests/golden/ hold the reference's own outputs
(tests/test_flash_attn.py: _softmax, _attention_forward, _flash_attention_forward)
and the PyTorch ops its tests compare against, on the shapes those tests use.
The reference computes in float32, so at N ~ 1000 its result is ~1e-5 from the
exact one; the oracle (float64) is held to 5e-5 of it, and the reference's own
test tolerance is 1e-4 (tests/test_flash_attn.py:151,168,186).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import ref_cpu

ATTN = [(2, 2), (1, 1), (2, 20), (2, 99), (20, 2), (99, 20), (1024, 64), (1023, 63), (1025, 65), (128, 64)]
SOFTMAX = [(255, 516), (1024, 1024)]
PIN_TOL = 5e-5


def test_manifest_records_pinning():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        man = json.load(f)
    assert set(man["attention"]) == {f"attn_{n}x{d}" for n, d in ATTN}
    for rec in man["attention"].values():
        assert rec["oracle_np_vs_ref_naive"] <= PIN_TOL
        assert rec["oracle_alg1_vs_ref_flash"] <= PIN_TOL
        assert rec["oracle_c_vs_ref_naive"] <= PIN_TOL
        # the reference's two paths agree with SDPA to its own tolerance
        assert rec["ref_naive_vs_sdpa"] < 1e-4 and rec["ref_flash_vs_sdpa"] < 1e-4
    for rec in man["softmax"].values():
        assert rec["oracle_np_vs_ref"] <= 1e-6 and rec["oracle_c_vs_ref"] <= 1e-6


@pytest.mark.parametrize("h,w", SOFTMAX)
def test_softmax_oracles_vs_golden(h, w, c_oracle):
    g = load_golden(f"softmax_{h}x{w}")
    x = g["x"]
    for out in (ref_cpu.softmax_rows(x), c_oracle.softmax_rows(x), c_oracle.softmax_rows(x, f32_arith=True)):
        assert np.abs(out - g["out_ref"]).max() <= 1e-6
        assert np.abs(out - g["out_torch"]).max() <= 1e-6
        assert np.allclose(out.sum(axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("n,d", ATTN)
def test_attention_oracles_vs_golden(n, d, c_oracle):
    g = load_golden(f"attn_{n}x{d}")
    q, k, v = g["q"], g["k"], g["v"]
    naive = ref_cpu.attention_naive(q, k, v)
    alg1, l, m = ref_cpu.attention_flash_alg1(q, k, v)
    fa2 = ref_cpu.attention_flash_fa2(q, k, v)
    cc = c_oracle.attention(q, k, v)
    for out in (naive, alg1, fa2, cc):
        assert out.shape == (n, d)
        assert np.abs(out - g["out_ref_naive"]).max() <= PIN_TOL
        assert np.abs(out - g["out_ref_flash"]).max() <= PIN_TOL
        assert np.abs(out - g["out_sdpa"]).max() <= PIN_TOL
    # the float64 restatements agree with each other far more tightly
    assert np.abs(naive - alg1).max() < 1e-12
    assert np.abs(naive - fa2).max() < 1e-12
    assert np.abs(naive - cc).max() < 1e-6          # C oracle rounds its result to float32
    # Algorithm-1 statistics: m + log(l) is the row log-sum-exp
    lse = ref_cpu.attention_lse(q, k)
    assert np.abs((m + np.log(l)) - lse).max() < 1e-10


@pytest.mark.parametrize("name", ["attn_causal_1x2x100x64", "attn_causal_2x1x257x128", "attn_causal_1x1x64x40"])
def test_causal_oracles_vs_sdpa_golden(name, c_oracle):
    """The causal mask has no counterpart in the reference; its oracles are pinned against the
    stored float64 output of torch's scaled_dot_product_attention(is_causal=True)."""
    g = load_golden(name)
    q, k, v, scale = g["q"], g["k"], g["v"], float(g["scale"])
    n = q.shape[-2]
    out_np = ref_cpu.attention_bhnd(q, k, v, scale=scale, causal=True)
    out_c = c_oracle.attention(q, k, v, scale=scale, causal_pos=np.arange(n))
    assert np.abs(out_np - g["out_sdpa_f64"]).max() <= 1e-6      # (the fixture is stored as float32)
    assert np.abs(out_c - g["out_sdpa_f64"]).max() <= 1e-6
    rows = np.array([0, n // 2, n - 1])
    out_s = c_oracle.attention(q[..., rows, :], k, v, scale=scale, causal_pos=rows)
    assert np.abs(out_s - out_np[..., rows, :]).max() <= 1e-6
    # row 0 sees key 0 only
    assert np.abs(out_np[..., 0, :] - v[..., 0, :]).max() <= 1e-7
    lse = ref_cpu.attention_lse(q, k, scale=scale, causal=True)
    assert np.abs(lse[..., 0] - scale * np.einsum("...d,...d->...", q[..., 0, :].astype(np.float64), k[..., 0, :].astype(np.float64))).max() < 1e-12


@pytest.mark.parametrize("name", ["attn_bwd_1x2x100x64", "attn_bwd_1x2x129x128_causal", "attn_bwd_2x1x64x64_causal"])
def test_backward_oracle_vs_autograd_golden(name):
    """The backward has no counterpart in the reference; its oracle is pinned against the stored
    float64 autograd gradients of torch's scaled_dot_product_attention, and checked here by
    central differences of the forward oracle as well."""
    g = load_golden(name)
    q, k, v, do = g["q"], g["k"], g["v"], g["dout"]
    scale, causal = float(g["scale"]), bool(g["causal"])
    dq, dk, dv = ref_cpu.attention_bwd_bhnd(q, k, v, do, scale=scale, causal=causal)
    for got, nm in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert np.abs(got - g[nm]).max() <= 1e-6 * max(1.0, np.abs(g[nm]).max())   # (fixture stored as float32)
    # directional derivative: <dout, (O(x + e t) - O(x - e t)) / 2e> = <grad, t>
    rng = np.random.default_rng(0)
    e = 1e-6
    for x, grad, idx in ((q, dq, 0), (k, dk, 1), (v, dv, 2)):
        t = rng.standard_normal(x.shape)
        args_p = [q.astype(np.float64), k.astype(np.float64), v.astype(np.float64)]
        args_m = [a.copy() for a in args_p]
        args_p[idx] = args_p[idx] + e * t
        args_m[idx] = args_m[idx] - e * t
        num = ((ref_cpu.attention_bhnd(*args_p, scale=scale, causal=causal)
                - ref_cpu.attention_bhnd(*args_m, scale=scale, causal=causal)) * do).sum() / (2 * e)
        assert abs(num - (grad * t).sum()) <= 1e-6 * max(1.0, abs(num))


def test_flash_block_sizes_match_reference_formula():
    # Bc = ceil(64 KiB / 4d), Br = min(Bc, d)  (tests/test_flash_attn.py:53-55)
    assert ref_cpu.flash_block_sizes(64) == (64, 256)
    assert ref_cpu.flash_block_sizes(128) == (128, 128)
    assert ref_cpu.flash_block_sizes(1) == (1, 16384)
    assert ref_cpu.flash_block_sizes(99) == (99, 166)


def test_bhnd_oracle_matches_per_head_naive():
    rng = np.random.default_rng(0)
    q, k, v = (rng.standard_normal((2, 3, 37, 16)).astype(np.float32) for _ in range(3))
    out = ref_cpu.attention_bhnd(q, k, v, scale=0.25)
    for b in range(2):
        for h in range(3):
            ref = ref_cpu.attention_naive(q[b, h], k[b, h], v[b, h], scale=0.25)
            assert np.abs(out[b, h] - ref).max() < 1e-12


def test_c_oracle_query_row_sample(c_oracle):
    rng = np.random.default_rng(1)
    q, k, v = (rng.standard_normal((2, 50, 24)).astype(np.float32) for _ in range(3))
    full = c_oracle.attention(q, k, v, scale=0.5)
    rows = np.array([0, 7, 49])
    part = c_oracle.attention(q[:, rows], k, v, scale=0.5)
    assert np.array_equal(part, full[:, rows])


SOFTMAX_DIM = ["softmax_dim_300x130_d0", "softmax_dim_4x37x96_d1", "softmax_dim_2x3x50x7_d2",
               "softmax_dim_3x700x5_d1", "softmax_dim_6x33x64_d2"]


@pytest.mark.parametrize("name", SOFTMAX_DIM)
def test_softmax_dim_oracle_vs_golden(name):
    """softmax along any dimension: the numpy oracle against the stored float64 torch.softmax,
    and against the row oracle on the tensor with that dimension moved last."""
    g = load_golden(name)
    x, dim = g["x"], int(g["dim"])
    out = ref_cpu.softmax_dim(x, dim)
    assert np.abs(out - g["out_torch_f64"]).max() <= 1e-7      # (the fixture is stored in float32)
    assert np.allclose(out.sum(axis=dim), 1.0, atol=1e-12)
    moved = np.moveaxis(x, dim, -1)
    rows = ref_cpu.softmax_rows(moved.reshape(-1, moved.shape[-1])).reshape(moved.shape)
    assert np.abs(np.moveaxis(rows, -1, dim) - out).max() <= 1e-15


SOFTMAX_FUSED = ["softmax_fused_0_2x3x40x40", "softmax_fused_1_2x2x33x33", "softmax_fused_2_3x50x50",
                 "softmax_fused_3_4x37x96", "softmax_fused_4_2x300x20", "softmax_fused_5_5x64"]


@pytest.mark.parametrize("name", SOFTMAX_FUSED)
def test_softmax_fused_oracle_vs_golden(name):
    """softmax(scale * x + mask) with -inf mask entries and the causal mask: the numpy oracle
    against the stored float64 torch.softmax of the composed input."""
    g = load_golden(name)
    mask = g["mask"] if bool(g["has_mask"]) else None
    out = ref_cpu.softmax_dim(g["x"], int(g["dim"]), scale=float(g["scale"]), mask=mask, causal=bool(g["causal"]))
    assert np.isfinite(out).all()
    assert np.abs(out - g["out_torch_f64"]).max() <= 1e-7
