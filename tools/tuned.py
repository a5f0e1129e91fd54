"""The tuning flavour of the library (libmyfa_gfx950_tuning.so) for tools/ and variant tests.

Same C ABI as the shipped library plus myfa_tuning_set / myfa_tuning_set_ptr (csrc/tuning.h).
`Tuned()` loads it beside the shipped one (separate code object, separate tuning record) and
offers the few calls the tools need with raw pointers, on torch's current stream.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch  # noqa: E402
from my_flash_attn_cuda import _lib  # noqa: E402

_TAG = {torch.float32: _lib.MYFA_F32, torch.float16: _lib.MYFA_F16, torch.bfloat16: _lib.MYFA_BF16}


class Tuned:
    def __init__(self, path=None):
        path = path or _lib.TUNING_LIB_PATH
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run `python flash-attention_amd/build.py`")
        self.lib = lib = ctypes.CDLL(path)
        vp, i64, f32, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
        lib.myfa_last_error.restype = ctypes.c_char_p
        lib.myfa_tuning_set.argtypes = [ctypes.c_char_p, ctypes.c_double]
        lib.myfa_tuning_set_ptr.argtypes = [ctypes.c_char_p, vp]
        lib.myfa_attn_fwd_bhnd_ex.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, ci, ci, vp]
        lib.myfa_matmul_f32.argtypes = [vp, vp, vp, i64, i64, i64, vp]
        lib.myfa_softmax_f32.argtypes = [vp, vp, i64, i64, vp]
        lib.myfa_attn_fwd_f32_ws_bytes.argtypes = [i64, i64]
        lib.myfa_attn_fwd_f32_ws_bytes.restype = i64
        lib.myfa_attn_fwd_f32_ws.argtypes = [vp, vp, vp, vp, i64, i64, f32, vp, i64, vp]
        lib.myfa_softmax_nd_ws_bytes.argtypes = [i64, i64, i64, ci]
        lib.myfa_softmax_nd_ws_bytes.restype = i64
        lib.myfa_softmax_nd_ws.argtypes = [vp, vp, i64, i64, i64, ci, f32, vp, i64, ci, vp, i64, vp]

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.myfa_last_error().decode("utf-8", "replace"))

    def set(self, **knobs):
        """Back to the defaults, then the given knobs (e.g. attn_variant=2, attn_thr=0)."""
        self._check(self.lib.myfa_tuning_set(b"reset", 0.0))
        for k, v in knobs.items():
            if k == "attn_dbg":
                self._check(self.lib.myfa_tuning_set_ptr(b"attn_dbg", ctypes.c_void_p(int(v))))
            else:
                self._check(self.lib.myfa_tuning_set(k.encode(), float(v)))

    def flash_attn_fwd(self, q, k, v, scale=1.0, causal=False, return_lse=False):
        B, H, N, d = q.shape
        out = torch.empty_like(q)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device) if return_lse else None
        self._check(self.lib.myfa_attn_fwd_bhnd_ex(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr() if return_lse else None,
            B, H, N, d, float(scale), _TAG[q.dtype], 1 if causal else 0, torch.cuda.current_stream().cuda_stream))
        return (out, lse) if return_lse else out

    def my_attention(self, q, kt, v, workspace=True):
        n, d = q.shape
        out = torch.empty_like(q)
        nbytes = self.lib.myfa_attn_fwd_f32_ws_bytes(n, d) if workspace else 0
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)
        self._check(self.lib.myfa_attn_fwd_f32_ws(q.data_ptr(), kt.data_ptr(), v.data_ptr(), out.data_ptr(), n, d, 1.0,
                                                  ws.data_ptr() if nbytes else None, nbytes, torch.cuda.current_stream().cuda_stream))
        return out

    def my_matmul(self, a, b):
        out = torch.empty((a.shape[0], b.shape[1]), dtype=a.dtype, device=a.device)
        self._check(self.lib.myfa_matmul_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1], b.shape[1],
                                             torch.cuda.current_stream().cuda_stream))
        return out

    def my_softmax(self, x):
        out = torch.empty_like(x)
        self._check(self.lib.myfa_softmax_f32(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                              torch.cuda.current_stream().cuda_stream))
        return out

    def softmax_dim0(self, x, out=None, ws=None):
        """softmax over dim 0 of a 2-D tensor; ws = a uint8 tensor to reuse, None = allocate what the library asks for.
        Returns (out, ws)."""
        out = torch.empty_like(x) if out is None else out
        ln, inner = x.shape
        nbytes = self.lib.myfa_softmax_nd_ws_bytes(1, ln, inner, _TAG[x.dtype])
        if ws is None and nbytes > 0:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        self._check(self.lib.myfa_softmax_nd_ws(x.data_ptr(), out.data_ptr(), 1, ln, inner, _TAG[x.dtype], 1.0, None, 1, 0,
                                                ws.data_ptr() if nbytes else None, nbytes, torch.cuda.current_stream().cuda_stream))
        return out, (ws if nbytes else None)
