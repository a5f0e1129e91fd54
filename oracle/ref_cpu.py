"""CPU oracle for the row-softmax / attention-forward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The shipped path
(``my_flash_attn_cuda`` -> C-ABI -> HIP kernels) never routes through here.

Every function is an independent numpy restatement of the algorithm the
reference states in Python, with the reference location it follows:

* ``softmax_rows``        <- tests/test_flash_attn.py:31-35  (``_softmax``)
* ``attention_naive``     <- tests/test_flash_attn.py:37-44  (``_attention_forward``)
                             and csrc/api.cpp:168-192 (``my_attention``: S=Q.K_t,
                             P=softmax(S), O=P.V, no scale, no mask)
* ``attention_flash_alg1``<- tests/test_flash_attn.py:46-101 (``_flash_attention_forward``,
                             Algorithm 1 of Dao et al. 2022: K/V blocks outer,
                             Q blocks inner, O renormalised every step)
* ``attention_bhnd``      <- the same naive statement applied per (batch, head),
                             with an explicit ``scale`` (reference semantics are
                             scale = 1.0, tests/test_flash_attn.py:165,183)

Pinning: ``oracle/make_golden.py`` checks these against the reference's own
Python functions (imported from the reference checkout, CPU, seed 1) and against
``torch.softmax`` / ``scaled_dot_product_attention(scale=1.0)`` on every shape
the reference tests use, and writes the fixtures in ``tests/golden/``.
"""
from math import ceil

import numpy as np


def _as(x, dtype):
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def softmax_rows(x, dtype=np.float64):
    """Numerically stable softmax over axis 1 of a 2-D array.

    exp(x - rowmax) / rowsum, the three steps of test_flash_attn.py:32-34.
    Computed in ``dtype`` (float64 by default: the checker is more precise
    than either implementation it judges).
    """
    x = _as(x, dtype)
    if x.ndim != 2:
        raise ValueError("softmax_rows expects a 2-D array")
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def softmax_dim(x, axis, dtype=np.float64, scale=1.0, mask=None, causal=False):
    """The same three steps along any one axis of an array of any rank: what
    torch.softmax(x, dim) computes; axis = -1 of a 2-D array is softmax_rows.

    With scale / mask / causal: softmax(scale * x + mask) -- the softmax step of the reference's
    attention composition (api.cpp:186-187: W = my_softmax(my_matmul_cublas(Q, K))) with the
    logit scale, an additive mask (numpy-broadcast against x from the right, -inf allowed) and
    the causal mask of (..., N, N) logits (element j of row i hidden when j > i; axis must be the
    last one) folded into the first step.  A slice that is -inf throughout is NaN, as in torch."""
    x = _as(x, dtype) * dtype(scale)
    if mask is not None:
        x = x + _as(mask, dtype)
    if causal:
        if axis not in (-1, x.ndim - 1) or x.ndim < 2 or x.shape[-1] != x.shape[-2]:
            raise ValueError("causal needs the last axis of (..., N, N) logits")
        n = x.shape[-1]
        x = np.where(np.arange(n)[None, :] > np.arange(n)[:, None], -np.inf, x)
    with np.errstate(invalid="ignore"):
        e = np.exp(x - x.max(axis=axis, keepdims=True))
        return e / e.sum(axis=axis, keepdims=True)


def attention_naive(q, k, v, scale=1.0, dtype=np.float64):
    """O = softmax(scale * Q K^T) V for 2-D Q, K, V of equal shape (N, d).

    Three materialised steps exactly as api.cpp:186-189 / test_flash_attn.py:40-43.
    ``k`` is in (N, d) layout here; the drop-in ``my_attention`` takes it
    pre-transposed, which callers of this oracle undo.
    """
    q, k, v = _as(q, dtype), _as(k, dtype), _as(v, dtype)
    if not (q.ndim == k.ndim == v.ndim == 2 and q.shape == k.shape == v.shape):
        raise ValueError("attention_naive expects three (N, d) arrays of one shape")
    s = (q @ k.T) * dtype(scale)
    return softmax_rows(s, dtype) @ v


def flash_block_sizes(d, sram_bytes=64 * 1024):
    """Block sizes of the reference's Algorithm-1 restatement.

    Bc = ceil(M / 4d) with the dummy M = 64 KiB, Br = min(Bc, d)
    (test_flash_attn.py:53-55; the same formula is sketched in
    csrc/my_flash_attn/flash_attention.cu:41-44).
    """
    bc = ceil(sram_bytes / (4 * d))
    return min(bc, d), bc


def attention_flash_alg1(q, k, v, dtype=np.float64, sram_bytes=64 * 1024):
    """Blockwise online-softmax attention in the FA-1 loop order.

    Follows test_flash_attn.py:57-99 step for step (outer loop over K/V
    blocks, inner loop over Q blocks, running rowmax m and rowsum l, O kept
    normalised and rewritten at every step).  The reference builds
    diag(l_new)^-1 with torch.inverse(diag_embed(.)) (:91); dividing each row
    by l_new is the same linear map and is what is done here.
    Returns (O, l, m) so the statistics can be checked too.
    """
    q, k, v = _as(q, dtype), _as(k, dtype), _as(v, dtype)
    n, d = q.shape
    br, bc = flash_block_sizes(d, sram_bytes)
    o = np.zeros((n, d), dtype)
    l = np.zeros((n,), dtype)
    m = np.full((n,), -np.inf, dtype)
    for j0 in range(0, n, bc):                       # :64  K/V block (outer)
        kj, vj = k[j0:j0 + bc], v[j0:j0 + bc]
        for i0 in range(0, n, br):                   # :70  Q block (inner)
            sl = slice(i0, i0 + br)
            s = q[sl] @ kj.T                         # :79
            m_blk = s.max(axis=1)                    # :82
            p = np.exp(s - m_blk[:, None])           # :83
            l_blk = p.sum(axis=1)                    # :84
            m_new = np.maximum(m[sl], m_blk)         # :87
            a_old = np.exp(m[sl] - m_new)
            a_blk = np.exp(m_blk - m_new)
            l_new = a_old * l[sl] + a_blk * l_blk    # :88
            o[sl] = ((l[sl] * a_old)[:, None] * o[sl]
                     + a_blk[:, None] * (p @ vj)) / l_new[:, None]   # :91-93
            m[sl], l[sl] = m_new, l_new              # :98-99
    return o, l, m


def attention_flash_fa2(q, k, v, scale=1.0, kv_block=64, dtype=np.float64):
    """The loop order the HIP kernels use: Q rows resident, K/V tiles streamed,
    un-normalised accumulator, one division by l at the end.

    Same online rescale as test_flash_attn.py:87-88, reordered; kept here so the
    tests can separate "the math of the reordering" from "the kernel".
    """
    q, k, v = _as(q, dtype), _as(k, dtype), _as(v, dtype)
    n, d = q.shape
    acc = np.zeros((n, d), dtype)
    l = np.zeros((n,), dtype)
    m = np.full((n,), -np.inf, dtype)
    for j0 in range(0, k.shape[0], kv_block):
        s = (q @ k[j0:j0 + kv_block].T) * dtype(scale)
        m_new = np.maximum(m, s.max(axis=1))
        alpha = np.exp(m - m_new)
        p = np.exp(s - m_new[:, None])
        l = alpha * l + p.sum(axis=1)
        acc = alpha[:, None] * acc + p @ v[j0:j0 + kv_block]
        m = m_new
    return acc / l[:, None]


def _causal_mask(s):
    """-inf above the diagonal of the last two axes: key j is hidden from query i when j > i
    (torch's scaled_dot_product_attention(is_causal=True); the reference has no mask)."""
    n, m = s.shape[-2:]
    s[..., np.triu(np.ones((n, m), dtype=bool), k=1)] = -np.inf
    return s


def attention_bhnd(q, k, v, scale=1.0, dtype=np.float64, causal=False):
    """Batched statement: O[b,h] = softmax(scale * Q[b,h] K[b,h]^T) V[b,h].

    q, k, v are (B, H, N, d) (or any leading dims).  Inputs are taken as given:
    for f16/bf16 parity the caller passes the *rounded* values widened to
    float, so the oracle judges the arithmetic and not the input rounding.
    """
    q, k, v = _as(q, dtype), _as(k, dtype), _as(v, dtype)
    s = np.einsum("...nd,...md->...nm", q, k) * dtype(scale)
    if causal:
        _causal_mask(s)
    s -= s.max(axis=-1, keepdims=True)
    np.exp(s, out=s)
    s /= s.sum(axis=-1, keepdims=True)
    return np.einsum("...nm,...md->...nd", s, v)


def attention_lse(q, k, scale=1.0, dtype=np.float64, causal=False):
    """Row log-sum-exp of scale * Q K^T (natural log), the statistic m + ln(l)
    that the reference's sketch keeps as two arrays
    (flash_attention.cu:37-39; test_flash_attn.py:58-59)."""
    q, k = _as(q, dtype), _as(k, dtype)
    s = np.einsum("...nd,...md->...nm", q, k) * dtype(scale)
    if causal:
        _causal_mask(s)
    mx = s.max(axis=-1)
    return mx + np.log(np.exp(s - mx[..., None]).sum(axis=-1))


def attention_bwd_bhnd(q, k, v, dout, scale=1.0, dtype=np.float64, causal=False):
    """Gradients (dq, dk, dv) of attention_bhnd with respect to its inputs for an upstream
    gradient `dout` of the output.  The reference has no backward (its fused forward is a
    stub, csrc/my_flash_attn/flash_attention.cu:23-69); this is the derivative of the forward
    statement above, pinned in oracle/make_golden.py against autograd of PyTorch's
    scaled_dot_product_attention in float64.

        P = softmax(scale Q K^T),  dV = P^T dO,  dP = dO V^T,
        dS = P o (dP - rowsum(dO o O)),  dQ = scale dS K,  dK = scale dS^T Q
    """
    q, k, v, dout = _as(q, dtype), _as(k, dtype), _as(v, dtype), _as(dout, dtype)
    s = np.einsum("...nd,...md->...nm", q, k) * dtype(scale)
    if causal:
        _causal_mask(s)
    s -= s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    o = np.einsum("...nm,...md->...nd", p, v)
    dv = np.einsum("...nm,...nd->...md", p, dout)
    dp = np.einsum("...nd,...md->...nm", dout, v)
    delta = (dout * o).sum(axis=-1, keepdims=True)
    ds = p * (dp - delta)
    dq = np.einsum("...nm,...md->...nd", ds, k) * dtype(scale)
    dk = np.einsum("...nm,...nd->...md", ds, q) * dtype(scale)
    return dq, dk, dv
