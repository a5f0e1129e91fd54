"""Column softmax (dim 0) of [len, W] for a sweep of shapes and of the workgroup count the row split
aims at (tuning flavour: softmax_split_wgs = -1 never split), on buffers rotated so that no launch
finds its input in a cache.  Each arm is checked against torch.softmax first."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned

def run(t, length, w, dtype, arms, iters=20):
    nbytes = length * w * torch.empty(0, dtype=dtype).element_size()
    nbuf = min(64, max(3, (1 << 30) // nbytes + 1))
    xs = [torch.randn(length, w, device="cuda").to(dtype) for _ in range(nbuf)]
    ref = torch.softmax(xs[0].double(), 0)
    out = torch.empty_like(xs[0])
    line = f"[{length},{w}] {str(dtype)[6:]:8s} {nbytes >> 20:5d} MiB "
    for arm in arms:
        t.set(softmax_split_wgs=arm)
        _, ws = t.softmax_dim0(xs[0], out)
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().max().item()
        for i in range(3):
            t.softmax_dim0(xs[i % nbuf], out, ws)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record()
        for i in range(iters):
            t.softmax_dim0(xs[i % nbuf], out, ws)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
        tm = ts[len(ts) // 2]
        line += f"| {arm:5d}: {tm * 1e3:7.1f} us {2 * nbytes / tm / 1e9:5.2f} TB/s err {err:.1e} "
    print(line, flush=True)

if __name__ == "__main__":
    t = Tuned()
    arms = (-1, 64, 128, 256, 512, 1024)
    for length, w in ((8192, 256), (8192, 1024), (8192, 4096), (16384, 4096), (2000, 8192), (100000, 512), (1000001, 24)):
        run(t, length, w, torch.float32, arms)
    run(t, 8192, 1024, torch.bfloat16, arms)
