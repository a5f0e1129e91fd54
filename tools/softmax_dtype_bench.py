import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch, my_flash_attn_cuda as mfa
for dt, es in ((torch.float32, 4), (torch.bfloat16, 2), (torch.float16, 2)):
    for n in (4096, 8192, 16384, 32768):
        nbuf = 3 if n * n * es <= (256 << 20) else 1
        xs = [torch.randn(n, n, device="cuda").to(dt) for _ in range(nbuf)]
        for i in range(3): mfa.softmax_rows(xs[i % nbuf])
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(ev):
            a.record(); mfa.softmax_rows(xs[i % nbuf]); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[10]
        print(f"{str(dt):16s} n={n:6d} {ms*1e3:9.1f} us  {2*n*n*es/ms/1e6:8.1f} GB/s  {n*n/ms/1e6:8.2f} Gelem/s", flush=True)
        del xs; torch.cuda.empty_cache()
