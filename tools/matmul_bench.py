import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch, my_flash_attn_cuda as mfa
def med(fn, it=20):
    for _ in range(3): fn()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    torch.cuda.synchronize()
    for a,b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
for n in (1024, 2048, 4096, 8192):
    a, b = torch.randn(n, n, device="cuda"), torch.randn(n, n, device="cuda")
    t2 = med(lambda: torch.matmul(a, b))
    line = f"n={n}: vendor {t2:8.3f} ms {2*n**3/t2/1e9:7.1f} TF"
    for blk in ("64", "128", ""):                       # "" = the library's own choice
        os.environ.pop("MYFA_MATMUL_BLOCK", None)
        if blk:
            os.environ["MYFA_MATMUL_BLOCK"] = blk
        t1 = med(lambda: mfa.my_matmul(a, b))
        line += f"   block {blk or 'auto':4s} {t1:8.3f} ms {2*n**3/t1/1e9:7.1f} TF (maxdiff {float((mfa.my_matmul(a,b)-torch.matmul(a,b)).abs().max()):.1e})"
    print(line, flush=True)
