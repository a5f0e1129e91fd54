"""A/B of two builds of the library on column softmax shapes, interleaved in one process:
python tools/ab_cols.py libA.so libB.so"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
libs = [Tuned(p) for p in sys.argv[1:]]
for (ln, w, dt) in ((8192, 8192, torch.float32), (16384, 16384, torch.float32), (8192, 8192, torch.bfloat16), (16384, 16384, torch.bfloat16),
                    (100000, 512, torch.float32), (8192, 1024, torch.bfloat16), (4096, 32768, torch.float16)):
    nb = ln * w * (4 if dt == torch.float32 else 2)
    nbuf = min(8, max(2, (1 << 30) // nb + 1))
    xs = [torch.randn(ln, w, device="cuda").to(dt) for _ in range(nbuf)]
    out = torch.empty_like(xs[0])
    res = [[] for _ in libs]
    wss = [None] * len(libs)
    ref = torch.softmax(xs[0][:, :64].double(), 0)
    for rnd in range(5):
        for li, t in enumerate(libs):
            _, wss[li] = t.softmax_dim0(xs[0], out, wss[li])
            if rnd == 0:
                torch.cuda.synchronize()
                err = (out[:, :64].double() - ref).abs().max().item()
            for i in range(2):
                t.softmax_dim0(xs[i % nbuf], out, wss[li])
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(ev):
                a.record(); t.softmax_dim0(xs[i % nbuf], out, wss[li]); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            res[li].append(ts[len(ts) // 2])
    for li, p in enumerate(sys.argv[1:]):
        r = sorted(res[li]); med = r[len(r) // 2]
        print(f"[{ln},{w}] {str(dt)[6:]:8s} {os.path.basename(p):14s} {med*1e3:8.1f} us  {2*nb/med/1e9:6.2f} TB/s  (err {err:.1e})", flush=True)
    del xs, out
    torch.cuda.empty_cache()
