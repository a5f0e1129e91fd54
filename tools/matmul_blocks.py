"""my_matmul at n^3 with either block kernel forced (tuning flavour) and the library's own choice."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
t = Tuned()
def med(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]
for n in (2048, 2560, 3072, 3584, 4096, 4608, 5120, 5632, 6144, 7168, 8192):
    a, b = torch.randn(n, n, device="cuda"), torch.randn(n, n, device="cuda")
    line = f"n={n}:"
    for blk in (0, 64, 128):
        t.set(matmul_block=blk)
        ms = med(lambda: t.my_matmul(a, b))
        line += f"  block {blk:3d}: {ms*1e3:8.1f} us {2*n**3/ms/1e9:6.1f} TF"
    ms = med(lambda: torch.matmul(a, b))
    print(line + f"  | vendor {ms*1e3:8.1f} us {2*n**3/ms/1e9:6.1f} TF", flush=True)
