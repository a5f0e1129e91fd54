#!/usr/bin/env python3
"""my_attention per call, the reference's way (back-to-back calls between two synchronisations), and
where the host's share goes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
import my_flash_attn_cuda as mfa
from my_flash_attn_cuda import _lib


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for n, d in ((128, 64), (1024, 64), (1025, 99), (2048, 128)):
    q, v = torch.randn(n, d, device="cuda"), torch.randn(n, d, device="cuda")
    kt = torch.randn(d, n, device="cuda")
    out = torch.empty_like(q)
    s = torch.cuda.current_stream().cuda_stream
    nb = _lib.lib.myfa_attn_fwd_f32_ws_bytes(n, d)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    print(f"N={n} d={d}: my_attention {per_call(lambda: mfa.my_attention(q, kt, v)):6.2f} us   "
          f"raw C call (fixed out, ws, stream) {per_call(lambda: _lib.lib.myfa_attn_fwd_f32_ws(q.data_ptr(), kt.data_ptr(), v.data_ptr(), out.data_ptr(), n, d, 1.0, ws.data_ptr(), nb, s)):6.2f} us   "
          f"no-split C call {per_call(lambda: _lib.lib.myfa_attn_fwd_f32(q.data_ptr(), kt.data_ptr(), v.data_ptr(), out.data_ptr(), n, d, 1.0, s)):6.2f} us   ws_bytes {nb}")
print(f"torch.empty_like {per_call(lambda: torch.empty_like(q)):5.2f} us")
