"""A/B of two builds of the library on the row softmax sweep sizes, interleaved in one process:
python tools/ab_softmax.py libA.so libB.so"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
libs = [Tuned(p) for p in sys.argv[1:]]
for n in (4096, 8192, 16384, 32768):
    nbuf = 3 if n <= 8192 else 1
    xs = [torch.randn(n, n, device="cuda") for _ in range(nbuf)]
    res = [[] for _ in libs]
    for rnd in range(7):
        for li, t in enumerate(libs):
            for i in range(3):
                t.my_softmax(xs[i % nbuf])
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(ev):
                a.record(); t.my_softmax(xs[i % nbuf]); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            res[li].append(ts[len(ts) // 2])
    for li, p in enumerate(sys.argv[1:]):
        r = sorted(res[li]); med = r[len(r) // 2]
        print(f"n={n} {os.path.basename(p):16s} {med*1e3:8.1f} us  {2*n*n*4/med/1e9:6.2f} TB/s", flush=True)
    del xs
    torch.cuda.empty_cache()
