"""Drop-in for the reference's ``my_flash_attn_cuda`` extension, on MI355X.

Same module name and the same five functions the reference registers with
pybind (csrc/api.cpp:194-200):

    my_softmax(inp, kernel_no)      my_attention(Q, K_t, V)
    my_matmul(A, B)   my_matmul_cublas(A, B)   my_matmul_out(out, A, B)

plus the batched forward the reference sketches but never wrote
(csrc/my_flash_attn/flash_attention.cu:23-69):

    flash_attn_fwd(q, k, v, scale=1.0, return_lse=False)     # (B, H, N, d)

The host side is Python: it validates arguments the way the reference's C++
does (CHECK_INPUT = device tensor + contiguous, csrc/my_flash_attn/util.hpp:7-9;
size checks, csrc/api.cpp:18,45,73,182-183; failures surface as RuntimeError,
which is what TORCH_CHECK raises through pybind), allocates the output with
torch, and calls the HIP kernels through the C ABI in ``include/myfa.h`` on
torch's *current* stream.  There is no CPU path: tensors must live on the GPU.
"""
import contextlib

import torch

from . import _lib
from ._lib import MYFA_BF16, MYFA_F16, MYFA_F32, device_info  # noqa: F401

__all__ = ["my_softmax", "my_attention", "my_matmul", "my_matmul_cublas", "my_matmul_out",
           "flash_attn_fwd", "flash_attn_bwd", "flash_attn_bwd_into", "flash_attn_func", "softmax_rows", "softmax", "device_info"]

_DTYPE_TAG = {torch.float32: MYFA_F32, torch.float16: MYFA_F16, torch.bfloat16: MYFA_BF16}


def _check_input(x, name):
    # CHECK_CUDA / CHECK_CONTIGUOUS (util.hpp:7-9)
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _same_device(x, ref, name):
    if x.device != ref.device:
        raise RuntimeError(f"{name} is on {x.device}, expected {ref.device}: all operands must be on one device")


def _check_float(x, name):
    # the reference reads data_ptr<float>(), which rejects every other dtype
    if x.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {x.dtype} for {name}")


# Host cost of a call matters for the small shapes of the reference's tests (a 1024 x 1024 softmax
# is 1.5 us of kernel): the raw handle of the current stream without building a Stream object, and
# no device switch when the tensor's device is the current one already.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_NO_SWITCH = contextlib.nullcontext()


def _stream():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _on(t):
    """Context that makes t's device current (the launch goes to that device's current stream)."""
    idx = t.device.index
    return _NO_SWITCH if idx == torch.cuda.current_device() else torch.cuda.device(idx)


def my_softmax(inp, kernel_no=1):
    """Row softmax of a 2-D float32 tensor; returns a new tensor.

    Mirrors my_softmax (csrc/api.cpp:110-166).  The reference's kernel_no picks
    one of five implementations of the same function (three of which its own
    checks reject, api.cpp:130,135,140); here 1..5 all run the one HIP kernel,
    and any other number raises like api.cpp:158.
    """
    _check_input(inp, "inp")
    _check_float(inp, "inp")
    if inp.dim() != 2:
        raise RuntimeError(f"my_softmax expects a 2-D tensor, got {inp.dim()}-D")
    if kernel_no not in (1, 2, 3, 4, 5):
        raise RuntimeError(f"Unsupported kernel number: {kernel_no}")
    h, w = inp.shape
    out = torch.empty_like(inp)   # every element is written; no zero-fill pass
    with _on(inp):
        if not _rows_with_workspace(inp, out, h, w):
            _lib.check(_lib.lib.myfa_softmax_f32(inp.data_ptr(), out.data_ptr(), h, w, _stream()))
    return out


def _rows_with_workspace(inp, out, h, w):
    """Few long rows (h far below the CU count, w beyond what a workgroup holds in registers): with a
    workspace (torch's caching allocator, stream-ordered) the library splits each row over workgroups.
    Returns False when the library would not split this shape."""
    tag = _DTYPE_TAG[inp.dtype]
    nbytes = _lib.lib.myfa_softmax_nd_ws_bytes(h, w, 1, tag)
    if nbytes <= 0:
        return False
    ws = torch.empty(nbytes, dtype=torch.uint8, device=inp.device)
    _lib.check(_lib.lib.myfa_softmax_nd_ws(inp.data_ptr(), out.data_ptr(), h, w, 1, tag, 1.0, None, 1, 0,
                                           ws.data_ptr(), nbytes, _stream()))
    return True


def softmax_rows(inp):
    """Row softmax of a 2-D float32, float16 or bfloat16 tensor; returns a new tensor of the
    same dtype.  New surface (the reference's my_softmax is float32 only, csrc/api.cpp:116-118):
    the arithmetic is float32 for every dtype, 16-bit rows only halve the bytes moved."""
    _check_input(inp, "inp")
    if inp.dtype not in _DTYPE_TAG:
        raise RuntimeError(f"unsupported dtype {inp.dtype}")
    if inp.dim() != 2:
        raise RuntimeError(f"softmax_rows expects a 2-D tensor, got {inp.dim()}-D")
    h, w = inp.shape
    out = torch.empty_like(inp)
    with _on(inp):
        if not _rows_with_workspace(inp, out, h, w):
            _lib.check(_lib.lib.myfa_softmax(inp.data_ptr(), out.data_ptr(), h, w, _DTYPE_TAG[inp.dtype], _stream()))
    return out


SOFTMAX_CAUSAL = 1   # MYFA_SOFTMAX_CAUSAL


def softmax(inp, dim=-1, scale=1.0, mask=None, causal=False):
    """torch.softmax(scale * inp + mask, dim) for a contiguous float32, float16 or bfloat16 tensor of
    any rank; returns a new tensor of the same dtype.  New surface (the reference's my_softmax is
    rows of a 2-D float32 matrix, csrc/api.cpp:110-166): the tensor is viewed as [outer, len, inner]
    around `dim`, the last dimension goes to the row kernels, any other to the column kernels.

    scale, mask and causal are fused into the load (no extra pass over the logits), for callers that
    keep the reference's attention as matmul -> softmax -> matmul (csrc/api.cpp:186-187):
      mask    float32 tensor whose shape equals the trailing dimensions of inp from some dimension
              <= dim on (e.g. (N, N) for (B, H, N, N) logits); it is broadcast over the leading ones
      causal  dim must be the last dimension and the last two dimensions equal: element j of row i
              is hidden when j > i, as scaled_dot_product_attention(is_causal=True) does
    """
    _check_input(inp, "inp")
    if inp.dtype not in _DTYPE_TAG:
        raise RuntimeError(f"unsupported dtype {inp.dtype}")
    if inp.dim() == 0:
        raise RuntimeError("softmax expects a tensor with at least one dimension")
    if not -inp.dim() <= dim < inp.dim():
        raise IndexError(f"Dimension out of range (expected to be in range of [{-inp.dim()}, {inp.dim() - 1}], but got {dim})")
    dim %= inp.dim()
    outer = inner = 1
    for n in inp.shape[:dim]:
        outer *= n
    for n in inp.shape[dim + 1:]:
        inner *= n
    out = torch.empty_like(inp)
    plain = mask is None and not causal and float(scale) == 1.0
    mask_outer = 1
    if mask is not None:
        _check_input(mask, "mask")
        _same_device(mask, inp, "mask")
        _check_float(mask, "mask")
        k = mask.dim()
        if k < inp.dim() - dim or k > inp.dim() or tuple(mask.shape) != tuple(inp.shape[inp.dim() - k:]):
            raise RuntimeError(f"mask shape {tuple(mask.shape)} must equal the trailing dimensions of the input "
                               f"{tuple(inp.shape)} from some dimension <= dim on")
        for n in inp.shape[inp.dim() - k:dim]:
            mask_outer *= n
    if causal and (dim != inp.dim() - 1 or inp.dim() < 2 or inp.shape[-1] != inp.shape[-2]):
        raise RuntimeError("causal=True needs dim to be the last dimension of (..., N, N) logits")
    with _on(inp):
        # long columns, or few long rows: with a workspace (torch's caching allocator, stream-ordered)
        # the library splits them over workgroups
        nbytes = _lib.lib.myfa_softmax_nd_ws_bytes(outer, inp.shape[dim], inner, _DTYPE_TAG[inp.dtype])
        if nbytes > 0:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=inp.device)
            _lib.check(_lib.lib.myfa_softmax_nd_ws(inp.data_ptr(), out.data_ptr(), outer, inp.shape[dim], inner,
                                                   _DTYPE_TAG[inp.dtype], float(scale),
                                                   mask.data_ptr() if mask is not None else None, mask_outer,
                                                   SOFTMAX_CAUSAL if causal else 0, ws.data_ptr(), nbytes, _stream()))
        elif plain:
            _lib.check(_lib.lib.myfa_softmax_nd(inp.data_ptr(), out.data_ptr(), outer, inp.shape[dim], inner,
                                                _DTYPE_TAG[inp.dtype], _stream()))
        else:
            _lib.check(_lib.lib.myfa_softmax_nd_ex(inp.data_ptr(), out.data_ptr(), outer, inp.shape[dim], inner,
                                                   _DTYPE_TAG[inp.dtype], float(scale),
                                                   mask.data_ptr() if mask is not None else None, mask_outer,
                                                   SOFTMAX_CAUSAL if causal else 0, _stream()))
    return out


# my_attention at the reference's sizes is a few microseconds of kernel: what the host spends per
# call shows.  The workspace size the library asks for depends on (n, d) only and is remembered; the
# workspace itself is kept per (device, stream) and grown when a call needs more -- calls on one
# stream are ordered, calls on different streams get different buffers -- so a call is one trip
# into the library and one allocation (the output).
_ATTN_WS_BYTES = {}
_ATTN_WS = {}


def _attn_workspace(n, d, device, stream):
    nbytes = _ATTN_WS_BYTES.get((n, d))
    if nbytes is None:
        nbytes = _ATTN_WS_BYTES[(n, d)] = int(_lib.lib.myfa_attn_fwd_f32_ws_bytes(n, d))
    if nbytes <= 0:
        return None, 0
    key = (device.index, stream)
    ws = _ATTN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if len(_ATTN_WS) > 64:      # streams come and go: do not hold buffers for all of them for ever
            _ATTN_WS.clear()
        ws = _ATTN_WS[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    return ws, nbytes


def my_attention(Q, K, V):
    """softmax(Q @ K) @ V for float32 Q:(n,d), K:(d,n) -- already transposed -- and V:(n,d).

    Mirrors my_attention (csrc/api.cpp:168-192): unscaled logits, no mask.  One
    fused kernel instead of SGEMM -> softmax -> SGEMM.
    """
    _check_input(Q, "Q"); _check_input(K, "K"); _check_input(V, "V")
    for t, name in ((Q, "Q"), (K, "K"), (V, "V")):
        _check_float(t, name)
        if t.dim() != 2:
            raise RuntimeError("Wrong sizes!")
    n, d = Q.shape
    if not (K.shape[0] == d and K.shape[1] == n and V.shape[0] == n and V.shape[1] == d):
        raise RuntimeError("Wrong sizes!")
    _same_device(K, Q, "K"); _same_device(V, Q, "V")
    out = torch.empty_like(Q)
    with _on(Q):
        # one head has too few query groups for the chip: with a workspace the library also splits
        # the keys over workgroups
        stream = _stream()
        ws, nbytes = _attn_workspace(n, d, Q.device, stream)
        if ws is not None:
            _lib.check(_lib.lib.myfa_attn_fwd_f32_ws(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(),
                                                     n, d, 1.0, ws.data_ptr(), nbytes, stream))
        else:
            _lib.check(_lib.lib.myfa_attn_fwd_f32(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(),
                                                  n, d, 1.0, stream))
    return out


ATTN_CAUSAL = 1   # MYFA_ATTN_CAUSAL


def flash_attn_fwd(q, k, v, scale=1.0, return_lse=False, causal=False):
    """Batched attention forward: softmax(scale * q @ k^T) @ v per (batch, head).

    q, k, v: contiguous (B, H, N, d) GPU tensors of one dtype (float16, bfloat16
    or float32); k is NOT transposed.  scale defaults to 1.0, the reference's
    semantics (its tests compare against scaled_dot_product_attention(scale=1.0),
    tests/test_flash_attn.py:165,183); pass d ** -0.5 for the usual scaling.
    float16/bfloat16 run on the matrix cores for every head dim that is a multiple of 8 up to 256
    (64 and 128 at full speed, the others on the next build up -- 64, 128 or 256 columns -- with zero
    columns); other head dims take a detour through padded or float32 copies (see below) and float32
    takes any d <= 512.
    With return_lse=True also returns the float32 (B, H, N) row log-sum-exp of the
    scaled logits (the m + log l statistic of tests/test_flash_attn.py:58-59,98-99).
    causal=True masks key j for query i when j > i, as
    scaled_dot_product_attention(is_causal=True) does (the reference has no mask).
    """
    _check_input(q, "q"); _check_input(k, "k"); _check_input(v, "v")
    _same_device(k, q, "k"); _same_device(v, q, "v")
    if q.dim() != 4:
        raise RuntimeError(f"flash_attn_fwd expects (B, H, N, d) tensors, got {q.dim()}-D")
    if not (q.shape == k.shape == v.shape):
        raise RuntimeError("Wrong sizes!")
    if not (q.dtype == k.dtype == v.dtype):
        raise RuntimeError("Scalar type mismatch!")
    if q.dtype not in _DTYPE_TAG:
        raise RuntimeError(f"unsupported dtype {q.dtype}")
    B, H, N, d = q.shape
    if q.dtype != torch.float32 and d > 0 and (d % 8 != 0 or d > 256):
        # Outside what the matrix-core kernels take (head dims that are multiples of 8 up to 256):
        # pad the head dim with zero columns up to the next multiple of 8 (the logits do not change,
        # the extra output columns are dropped), or, beyond 256, run the float32 kernel (any d <= 512)
        # on widened copies.  Both cost extra passes over q, k, v; neither changes the result's dtype.
        if d <= 256:
            pad = (-d) % 8
            qp, kp, vp = (torch.nn.functional.pad(t, (0, pad)) for t in (q, k, v))
            res = flash_attn_fwd(qp, kp, vp, scale=scale, return_lse=return_lse, causal=causal)
            return (res[0][..., :d].contiguous(), res[1]) if return_lse else res[..., :d].contiguous()
        res = flash_attn_fwd(q.float(), k.float(), v.float(), scale=scale, return_lse=return_lse, causal=causal)
        return (res[0].to(q.dtype), res[1]) if return_lse else res.to(q.dtype)
    out = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device) if return_lse else None
    with _on(q):
        _lib.check(_lib.lib.myfa_attn_fwd_bhnd_ex(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
            lse.data_ptr() if return_lse else None,
            B, H, N, d, float(scale), _DTYPE_TAG[q.dtype], ATTN_CAUSAL if causal else 0, _stream()))
    return (out, lse) if return_lse else out


def attn_fwd_kernel_name(d, dtype, causal=False):
    """Name of the device kernel flash_attn_fwd launches for this head dim and dtype (what a kernel
    trace shows), or None where the C entry point would refuse the call."""
    name = _lib.lib.myfa_attn_fwd_bhnd_kernel(int(d), _DTYPE_TAG[dtype], ATTN_CAUSAL if causal else 0)
    return name.decode() if name else None


BWD_MAX_HEAD_DIM = 256                       # float16 / bfloat16
BWD_MAX_HEAD_DIM_F32 = 512                   # float32: the float32 forward's whole range
BWD_NATIVE_HEAD_DIMS = (64, 128, 192, 256)   # what the 16-bit backward kernels are built for


def _bwd_max_head_dim(dtype, native=False):
    """Widest head dim of the backward: 16-bit tensors beyond 256 columns take the float32 kernels through
    float32 copies (as their forward does), which an entry point that may not allocate cannot."""
    return BWD_MAX_HEAD_DIM if native and dtype != torch.float32 else BWD_MAX_HEAD_DIM_F32


def flash_attn_bwd(dout, q, k, v, out, lse, scale=1.0, causal=False):
    """Gradients (dq, dk, dv) of flash_attn_fwd for float16 / bfloat16 / float32 (B, H, N, d)
    tensors: 16-bit with d <= 256 (64, 128, 192 and 256 natively, the others through copies zero-padded
    to the next of those) and, like their forward, through float32 copies up to 512; float32 with any
    d <= 512 (natively).  `out` and `lse` are the forward's results
    (return_lse=True), `dout` the gradient of `out`.  The reference has no backward; this is the derivative of the forward
    it sketches (csrc/my_flash_attn/flash_attention.cu:23-69)."""
    for t, name in ((dout, "dout"), (q, "q"), (k, "k"), (v, "v"), (out, "out"), (lse, "lse")):
        _check_input(t, name)
        _same_device(t, q, name)
    if q.dim() != 4 or not (q.shape == k.shape == v.shape == out.shape == dout.shape):
        raise RuntimeError("Wrong sizes!")
    if not (q.dtype == k.dtype == v.dtype == out.dtype == dout.dtype):
        raise RuntimeError("Scalar type mismatch!")
    if q.dtype not in _DTYPE_TAG:
        raise RuntimeError(f"flash_attn_bwd supports float16, bfloat16 and float32, got {q.dtype}")
    B, H, N, d = q.shape
    if d > _bwd_max_head_dim(q.dtype):
        raise RuntimeError(f"flash_attn_bwd supports head dims up to {_bwd_max_head_dim(q.dtype)} for {q.dtype}, got {d}")
    if lse.dtype != torch.float32 or tuple(lse.shape) != (B, H, N):
        raise RuntimeError("lse must be the float32 (B, H, N) tensor returned by flash_attn_fwd")
    if q.dtype != torch.float32 and d > BWD_MAX_HEAD_DIM:
        # the forward of these shapes ran the float32 kernel on float32 copies; so does the backward
        grads = flash_attn_bwd(dout.float(), q.float(), k.float(), v.float(), out.float(), lse, scale=scale, causal=causal)
        return tuple(g.to(q.dtype) for g in grads)
    if q.dtype != torch.float32 and d not in BWD_NATIVE_HEAD_DIMS and d > 0:
        # zero columns up to the next head dim the kernels are built for change nothing
        pad = min(n for n in BWD_NATIVE_HEAD_DIMS if n > d) - d
        padded = [torch.nn.functional.pad(t, (0, pad)) for t in (dout, q, k, v, out)]
        return tuple(g[..., :d].contiguous() for g in flash_attn_bwd(*padded, lse, scale=scale, causal=causal))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty((2, B, H, N), dtype=torch.float32, device=q.device)   # workspace: -rowsum(dout * out), lse in base 2
    with _on(q):
        _lib.check(_lib.lib.myfa_attn_bwd_bhnd(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(),
            B, H, N, d, float(scale), _DTYPE_TAG[q.dtype], ATTN_CAUSAL if causal else 0, _stream()))
    return dq, dk, dv


BWD_STATIC_WS_ROWS = 1048576   # MYFA_BWD_STATIC_WS_ROWS


def flash_attn_bwd_into(dout, q, k, v, out, lse, dq, dk, dv, scale=1.0, causal=False):
    """flash_attn_bwd for callers that may not allocate: the gradients go into the caller's
    dq, dk, dv and the workspace is the library's own device array (B*H*N <= BWD_STATIC_WS_ROWS;
    calls that overlap on one GPU race on it).  Results equal flash_attn_bwd's bit for bit."""
    for t, name in ((dout, "dout"), (q, "q"), (k, "k"), (v, "v"), (out, "out"), (lse, "lse"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _check_input(t, name)
        _same_device(t, q, name)
    if q.dim() != 4 or not (q.shape == k.shape == v.shape == out.shape == dout.shape == dq.shape == dk.shape == dv.shape):
        raise RuntimeError("Wrong sizes!")
    if not (q.dtype == k.dtype == v.dtype == out.dtype == dout.dtype == dq.dtype == dk.dtype == dv.dtype):
        raise RuntimeError("Scalar type mismatch!")
    if q.dtype not in _DTYPE_TAG:
        raise RuntimeError(f"flash_attn_bwd supports float16, bfloat16 and float32, got {q.dtype}")
    B, H, N, d = q.shape
    if d > _bwd_max_head_dim(q.dtype, native=True) or (q.dtype != torch.float32 and d not in BWD_NATIVE_HEAD_DIMS):
        raise RuntimeError(f"flash_attn_bwd_into supports head dims 64, 128, 192 and 256 (float32: any up to 512), got {d} "
                           "(flash_attn_bwd pads the others through copies, which this entry may not make)")
    if lse.dtype != torch.float32 or tuple(lse.shape) != (B, H, N):
        raise RuntimeError("lse must be the float32 (B, H, N) tensor returned by flash_attn_fwd")
    with _on(q):
        _lib.check(_lib.lib.myfa_attn_bwd_bhnd_static_ws(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
            B, H, N, d, float(scale), _DTYPE_TAG[q.dtype], ATTN_CAUSAL if causal else 0, _stream()))
    return dq, dk, dv


class _FlashAttnFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale, causal):
        if q.dtype not in _DTYPE_TAG or q.shape[-1] > _bwd_max_head_dim(q.dtype):
            raise RuntimeError(f"flash_attn_func is differentiable for float16 / bfloat16 / float32 and head dims up to "
                               f"{BWD_MAX_HEAD_DIM_F32}; got {q.dtype}, d = {q.shape[-1]}")
        out, lse = flash_attn_fwd(q, k, v, scale=scale, return_lse=True, causal=causal)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale, ctx.causal = scale, causal
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = flash_attn_bwd(dout.contiguous(), q, k, v, out, lse, scale=ctx.scale, causal=ctx.causal)
        return dq, dk, dv, None, None


def flash_attn_func(q, k, v, scale=1.0, causal=False):
    """flash_attn_fwd with autograd support: float16 / bfloat16 / float32 with head dims up to 512,
    everything flash_attn_fwd serves (16-bit: 64, 128, 192 and 256 natively, the others through zero-padded
    copies, beyond 256 through float32 copies, in both directions; float32: any, natively).  Anything else
    is refused here, before any work."""
    return _FlashAttnFunc.apply(q, k, v, scale, causal)


# ---- the matmul trio ---------------------------------------------------------
# Not on the softmax/attention path.  my_matmul / my_matmul_out run the library's own
# float32 MFMA kernel, as the reference's run its own 16x16-tile kernel
# (csrc/my_flash_attn/matmul.cu:22-50, csrc/api.cpp:13-66); my_matmul_cublas
# (csrc/api.cpp:68-105) is the vendor GEMM of this platform (torch.matmul), as the
# reference's is the vendor GEMM of its platform.

def _matmul_checks(A, B):
    _check_input(A, "A"); _check_input(B, "B")
    _same_device(B, A, "B")
    if A.dim() != 2 or B.dim() != 2 or A.shape[1] != B.shape[0]:
        raise RuntimeError("Size mismatch!")


def _matmul_into(out, A, B):
    _check_float(A, "A"); _check_float(B, "B"); _check_float(out, "out")
    m, k = A.shape
    n = B.shape[1]
    with _on(A):
        _lib.check(_lib.lib.myfa_matmul_f32(A.data_ptr(), B.data_ptr(), out.data_ptr(), m, k, n, _stream()))
    return out


def my_matmul(A, B):
    _matmul_checks(A, B)
    return _matmul_into(torch.empty((A.shape[0], B.shape[1]), dtype=A.dtype, device=A.device), A, B)


def my_matmul_cublas(A, B):
    _matmul_checks(A, B)
    _check_float(A, "A"); _check_float(B, "B")   # the reference reads data_ptr<float>() (api.cpp:96-99)
    return torch.matmul(A, B)


def my_matmul_out(out, A, B):
    _check_input(out, "out")
    _matmul_checks(A, B)
    _same_device(out, A, "out")
    if tuple(out.shape) != (A.shape[0], B.shape[1]):
        raise RuntimeError("Size mismatch!")
    if not (out.dtype == A.dtype == B.dtype):
        raise RuntimeError("Scalar type mismatch!")
    # `out` is written while A and B are being read: an output that shares storage with an input
    # is refused (the C entry point checks the address ranges and its message surfaces here)
    return _matmul_into(out, A, B)
