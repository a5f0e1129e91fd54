#!/usr/bin/env python3
"""Diagnostic: attention-kernel time against the DATA it is fed (same shape, same code).

Switching activity in the MFMA datapath sets the power drawn, and the power limit sets the
clock, so all-zero or low-entropy operands run faster than random ones when the kernel is
power-bound rather than issue-bound.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
import my_flash_attn_cuda as mfa

B, H, N, d = 4, 32, 4096, 128
flops = 4.0 * B * H * N * N * d
torch.manual_seed(1)


def timed(q, k, v, iters=20):
    for _ in range(3):
        mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


mk = lambda f: tuple(f().to(torch.bfloat16).cuda() for _ in range(3))
cases = {
    "randn": mk(lambda: torch.randn(B, H, N, d)),
    "zeros": mk(lambda: torch.zeros(B, H, N, d)),
    "ones": mk(lambda: torch.ones(B, H, N, d)),
    "randn*1e-3": mk(lambda: torch.randn(B, H, N, d) * 1e-3),
    "sign(randn)": mk(lambda: torch.sign(torch.randn(B, H, N, d))),
}
for rnd in range(2):
    for name, (q, k, v) in cases.items():
        ms = timed(q, k, v)
        print(f"{name:14s} {ms:8.4f} ms  {flops / ms / 1e9:8.1f} TF", flush=True)

# Calibration: the vendor GEMM (hipBLASLt through torch.matmul) on the same box, same dtypes:
# what a plain bf16 MFMA GEMM sustains under the same power limit, per operand data.
n = 8192
gflops = 2.0 * n ** 3
for name, f in {"randn": lambda: torch.randn(n, n), "zeros": lambda: torch.zeros(n, n), "ones": lambda: torch.ones(n, n)}.items():
    a, b = f().to(torch.bfloat16).cuda(), f().to(torch.bfloat16).cuda()
    for _ in range(3):
        torch.matmul(a, b)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    torch.cuda.synchronize()
    for x, y in ev:
        x.record(); torch.matmul(a, b); y.record()
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)[10]
    print(f"matmul bf16 {n}^3 {name:6s} {ms:8.4f} ms  {gflops / ms / 1e9:8.1f} TF", flush=True)
