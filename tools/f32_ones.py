#!/usr/bin/env python3
"""Batched float32 attention forward on random and on all-ones operands (the clock the chip
holds differs: all-ones shows the kernel's cycle efficiency)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
import my_flash_attn_cuda as mfa

B, H, N, d = 4, 32, 4096, 128
for name, make in (("randn", torch.randn), ("ones", torch.ones)):
    q, k, v = (make(B, H, N, d, device="cuda") for _ in range(3))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    print(f"{name}: {ms:.3f} ms {4 * B * H * N * N * d / ms / 1e9:.2f} TFLOP/s", flush=True)
