#!/usr/bin/env python3
"""Where the microseconds of a small my_softmax call go on the host (1000 back-to-back calls each)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
import my_flash_attn_cuda as mfa
from my_flash_attn_cuda import _lib

x = torch.randn(1024, 1024, device="cuda")
out = torch.empty_like(x)


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


s = torch.cuda.current_stream().cuda_stream
print(f"my_softmax                      {per_call(lambda: mfa.my_softmax(x)):6.2f} us")
print(f"torch.softmax                   {per_call(lambda: torch.softmax(x, 1)):6.2f} us")
print(f"raw C call, fixed out/stream    {per_call(lambda: _lib.lib.myfa_softmax_f32(x.data_ptr(), out.data_ptr(), 1024, 1024, s)):6.2f} us")
print(f"torch.empty_like                {per_call(lambda: torch.empty_like(x)):6.2f} us")
print(f"current_stream().cuda_stream    {per_call(lambda: torch.cuda.current_stream().cuda_stream):6.2f} us")


def ctx():
    with torch.cuda.device(x.device):
        pass


print(f"with torch.cuda.device(dev)     {per_call(ctx):6.2f} us")
print(f"current_device()                {per_call(lambda: torch.cuda.current_device()):6.2f} us")
print(f"x.device.index                  {per_call(lambda: x.device.index):6.2f} us")
print(f"is_cuda+is_contiguous+dtype+dim {per_call(lambda: (x.is_cuda, x.is_contiguous(), x.dtype, x.dim(), x.shape)):6.2f} us")
