"""my_matmul on small problems with k in one piece or split over two workgroups (tuning flavour)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
t = Tuned()
def med(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]
for (m, k, n) in ((256, 256, 256), (512, 512, 512), (1024, 1024, 1024), (1024, 4096, 1024), (768, 2048, 768), (1536, 1536, 1536), (128, 8192, 128), (2048, 2048, 2048), (1000, 1111, 1000)):
    a, b = torch.randn(m, k, device="cuda"), torch.randn(k, n, device="cuda")
    ref = a.double() @ b.double()
    line = f"{m}x{k}x{n}:"
    for ks in (1, 2, 0):
        t.set(matmul_ksplit=ks)
        out = t.my_matmul(a, b)
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        ms = med(lambda: t.my_matmul(a, b))
        line += f"  ksplit {ks}: {ms*1e3:7.1f} us {2*m*k*n/ms/1e9:6.1f} TF (rel err {err:.1e})"
    ms = med(lambda: torch.matmul(a, b))
    print(line + f"  | vendor {ms*1e3:7.1f} us", flush=True)
