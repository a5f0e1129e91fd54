"""my_matmul at 4096^3 / 8192^3 through the shipped library (Python entry point) and through the
tuning flavour, interleaved in one process."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
import my_flash_attn_cuda as mfa
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
for n in (4096, 8192, 4096):
    a, b = torch.randn(n, n, device="cuda"), torch.randn(n, n, device="cuda")
    for rnd in range(2):
        ms1 = med(lambda: mfa.my_matmul(a, b)); ms2 = med(lambda: t.my_matmul(a, b)); ms3 = med(lambda: torch.matmul(a, b))
        print(f"n={n}: shipped {ms1*1e3:8.1f} us {2*n**3/ms1/1e9:6.1f} TF | tuned {ms2*1e3:8.1f} us {2*n**3/ms2/1e9:6.1f} TF | vendor {2*n**3/ms3/1e9:6.1f} TF", flush=True)
