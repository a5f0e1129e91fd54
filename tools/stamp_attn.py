#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the ping-pong attention kernel (stamped build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned
import torch
mfa = Tuned()
args = dict(kv.split("=") for kv in sys.argv[1:])
B, H, N, d = map(int, args.pop("shape", "4,32,4096,128").split(","))
torch.manual_seed(1)
q, k, v = (torch.randn(B, H, N, d, dtype=torch.bfloat16, device="cuda") for _ in range(3))
nwg = B * H * (N // 256)
dbg = torch.zeros(nwg * 24, dtype=torch.int64, device="cuda")
causal = bool(int(args.pop("causal", "0")))
mfa.set(attn_dbg=dbg.data_ptr(), **{key: float(val) for key, val in args.items()})
for _ in range(3):
    mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5, causal=causal)
torch.cuda.synchronize()
x = dbg.view(nwg, 2, 12).double().cpu()
if causal:
    # blockIdx -> logical id (XCD-chunked) -> q-block (reversed within the head), as in the kernel
    import numpy as np
    nq = N // 256
    bid = np.arange(nwg); xcd = bid & 7; idx = bid >> 3; qq, rr = nwg >> 3, nwg & 7
    lid = np.where(xcd < rr, xcd * (qq + 1), rr * (qq + 1) + (xcd - rr) * qq) + idx
    qblk = nq - 1 - (lid % nq)
    xs = x.numpy()
    loop = xs[:, 0, 0] + xs[:, 0, 1] + xs[:, 0, 2] + xs[:, 0, 3]
    print("qblk tiles  loop/tile  prologue  epilogue   total   (wave 0, median over heads)")
    tot = 0.0
    for b in range(nq):
        sel = qblk == b
        tiles = 4 * (b + 1)
        lp, pro, epi = np.median(loop[sel]), np.median(xs[sel, 0, 5]), np.median(xs[sel, 0, 6])
        print(f"{b:4d} {tiles:5d} {lp / tiles:10.0f} {pro:9.0f} {epi:9.0f} {lp + pro + epi:8.0f}")
        tot += (lp + pro + epi) * sel.sum()
    print(f"sum over workgroups of (prologue + loop + epilogue) cycles / 256 CUs: {tot / 256:.0f}")
    sys.exit(0)
T = N // 64
names = ["M body", "M barrier wait", "X body", "X barrier wait", "M mfma part", "prologue/64", "epilogue/64"]
pro = ["to fetch issue", "fetch wait", "first barrier", "QK(0)", "X(0)+barrier(s)"]
for g in range(2):
    print(f"wave {4*g}: " + "  ".join(f"{n}: {x[:, g, i].median().item() / T:8.1f}" for i, n in enumerate(names)) + "  (cycles per tile, median over WGs)")
for g in range(2):
    print(f"wave {4*g} prologue cycles (median over WGs): " + "  ".join(f"{n}: {x[:, g, 7 + i].median().item():8.0f}" for i, n in enumerate(pro))
          + f"  total: {x[:, g, 5].median().item():8.0f}   epilogue: {x[:, g, 6].median().item():8.0f}")
first = x[:min(256, nwg // 2), 0, 5].median().item(); later = x[min(256, nwg // 2):, 0, 5].median().item()
print(f"prologue of the first 256 workgroups: {first:.0f}; of the rest: {later:.0f}")
