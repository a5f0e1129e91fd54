"""Soak of the workspace-split softmax kernels (cross-workgroup exchange inside a launch): many
launches on fresh random data; each result must equal a second split launch bit for bit (the merge
order is fixed, whichever workgroup arrives last) and the unsplit form of the same library to
rounding of the longer sums."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
from tuned import Tuned
t = Tuned()
torch.manual_seed(0)
bad = 0
shapes = [(8192, 256), (100000, 64), (4100, 1024), (8192, 1024), (3000, 4096)]
for it in range(400):
    ln, w = shapes[it % len(shapes)]
    dt = (torch.float32, torch.bfloat16)[(it // len(shapes)) % 2]
    x = (torch.randn(ln, w, device="cuda") * 4).to(dt)
    t.set(softmax_split_wgs=0)
    a, ws = t.softmax_dim0(x)
    a2, _ = t.softmax_dim0(x, None, ws)
    t.set(softmax_split_wgs=-1)
    b, _ = t.softmax_dim0(x)
    d = (a.float() - b.float()).abs().max().item()
    if not torch.equal(a, a2) or not d <= (1e-5 if dt == torch.float32 else 4e-3) or ws is None:
        bad += 1
        print("MISMATCH", it, ln, w, dt, d, flush=True)
print("soak done, mismatches:", bad, flush=True)
