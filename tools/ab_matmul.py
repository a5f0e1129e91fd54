"""A/B of two builds of the library on my_matmul, interleaved in one process:
python tools/ab_matmul.py libA.so libB.so"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
libs = [Tuned(p) for p in sys.argv[1:]]
def med(fn, iters=15, warm=4):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]
for (m, k, n) in ((256, 256, 256), (512, 512, 512), (1024, 1024, 1024), (1024, 4096, 1024), (1536, 1536, 1536), (2048, 2048, 2048), (3072, 3072, 3072),
                  (3584, 3584, 3584), (1000, 1111, 1000), (127, 515, 33), (4096, 64, 4096)):
    a, b = torch.randn(m, k, device="cuda"), torch.randn(k, n, device="cuda")
    ref = a.double() @ b.double()
    line = f"{m}x{k}x{n}:"
    for t, p in zip(libs, sys.argv[1:]):
        out = t.my_matmul(a, b)
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        ms = min(med(lambda: t.my_matmul(a, b)) for _ in range(3))
        line += f"  {os.path.basename(p)}: {ms*1e3:7.1f} us {2*m*k*n/ms/1e9:6.1f} TF (err {err:.0e})"
    print(line, flush=True)
