"""Forward at head dims above 128 (the 256-column build) against torch SDPA and the float32 route."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flash-attention_amd"))
import my_flash_attn_cuda as m

def med(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

for (B, H, N, d) in ((4, 16, 4096, 256), (4, 16, 4096, 192), (2, 16, 8192, 256), (8, 16, 1024, 256)):
    for causal in (False, True):
        q, k, v = (torch.randn(B, H, N, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        sc = d ** -0.5
        t = med(lambda: m.flash_attn_fwd(q, k, v, scale=sc, causal=causal))
        t2 = med(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=sc))
        fl = 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)
        err = (m.flash_attn_fwd(q, k, v, scale=sc, causal=causal).float() - torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=sc).float()).abs().max().item()
        print(f"B{B} H{H} N{N} d{d} causal={int(causal)}: {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF | torch SDPA {t2*1e3:8.1f} us {fl/t2/1e9:7.1f} TF | maxdiff {err:.2e}", flush=True)
