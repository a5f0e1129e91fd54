"""my_attention at the reference's largest test shape (N=1024, d=64) and at config 1 (N=128, d=64),
300 calls each, for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/my_attention_trace.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flash-attention_amd"))
import my_flash_attn_cuda as m
for n, d in ((1024, 64), (128, 64), (1025, 99)):
    q, kt, v = torch.randn(n, d, device="cuda"), torch.randn(d, n, device="cuda"), torch.randn(n, d, device="cuda")
    for _ in range(20):
        m.my_attention(q, kt, v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        m.my_attention(q, kt, v)
    torch.cuda.synchronize()
    print(f"N={n} d={d}: {(time.perf_counter() - t0) / 300 * 1e6:.2f} us per call (300 back to back)", flush=True)
