#!/usr/bin/env python3
"""Times the attention backward (three kernels) next to the forward; 2.5x the forward's
algorithmic flops (five N x N x d products against two)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
import my_flash_attn_cuda as mfa

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="4,32,4096,128")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--causal", action="store_true")
ap.add_argument("--no-torch", action="store_true", help="skip the PyTorch SDPA backward (profiling runs)")
ap.add_argument("--fill", default="randn", choices=["randn", "ones"], help="ones: operands that toggle few bits (the power limit then binds less)")
a = ap.parse_args()
B, H, N, d = map(int, a.shape.split(","))
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
torch.manual_seed(1)
q, k, v, do = (torch.randn(B, H, N, d, dtype=dt, device="cuda") for _ in range(4))
if a.fill == "ones":
    q, k, v, do = (torch.ones_like(x) for x in (q, k, v, do))
scale = d ** -0.5
out, lse = mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True, causal=a.causal)


def med(fn, it=15):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    torch.cuda.synchronize()
    for x, y in ev:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in ev)[it // 2]


half = 0.5 if a.causal else 1.0
f = 4.0 * B * H * N * N * d * half
t_f = med(lambda: mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True, causal=a.causal))
t_b = med(lambda: mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=scale, causal=a.causal))
print(f"forward  {t_f:8.3f} ms  {f / t_f / 1e9:8.1f} TFLOP/s")
print(f"backward {t_b:8.3f} ms  {2.5 * f / t_b / 1e9:8.1f} TFLOP/s (algorithmic 2.5x forward)")
if a.no_torch:
    sys.exit(0)
qf, kf, vf = (x.detach().clone().requires_grad_(True) for x in (q, k, v))
o = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, is_causal=a.causal, scale=scale)
t_t = med(lambda: torch.autograd.grad(o, (qf, kf, vf), do, retain_graph=True), it=7)
print(f"torch SDPA backward {t_t:8.3f} ms  {2.5 * f / t_t / 1e9:8.1f} TFLOP/s")
