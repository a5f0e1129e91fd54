#!/usr/bin/env python3
"""Times the float32 backward (delta + dK/dV + dQ launches) against the float32 matrix peak and
against PyTorch's float32 scaled_dot_product_attention backward on the same tensors.

    python tools/bench_bwd_f32.py [B H N d]...      (default: a few shapes)
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention_amd"))
import torch

import my_flash_attn_cuda as mfa

F32_MFMA_PEAK = 157.3   # TFLOP/s, v_mfma_f32_16x16x4_f32 on 256 CUs at 2.4 GHz


def median_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    args = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(args[i:i + 4]) for i in range(0, len(args), 4)] or [(4, 32, 4096, 128), (8, 16, 1024, 64), (4, 16, 2048, 256), (2, 8, 4096, 32)]
    for causal in (False, True):
        for B, H, N, d in shapes:
            torch.manual_seed(0)
            q, k, v, do = (torch.randn(B, H, N, d, device="cuda") for _ in range(4))
            scale = d ** -0.5
            out, lse = mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True, causal=causal)
            ms = median_ms(lambda: mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=scale, causal=causal))
            flops = 2.5 * 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)     # five products counted; seven are executed
            qf, kf, vf = (x.detach().clone().requires_grad_(True) for x in (q, k, v))
            o = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, scale=scale, is_causal=causal)
            ms_t = median_ms(lambda: torch.autograd.grad(o, (qf, kf, vf), do, retain_graph=True), iters=5, warm=2)
            tf = flops / ms / 1e9
            print(f"B{B} H{H} N{N} d{d} causal={int(causal)}: {ms:8.3f} ms  {tf:6.1f} TFLOP/s algorithmic "
                  f"({tf / F32_MFMA_PEAK:.3f} of the f32 matrix peak; executed x1.4)   torch sdpa f32 backward {ms_t:8.3f} ms", flush=True)
            del q, k, v, do, out, lse, qf, kf, vf, o
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
