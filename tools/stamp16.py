#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the 16x16x32 attention kernel (stamped build, tuning flavour).

    python tools/stamp16.py [shape=4,32,4096,128] [data=randn|ones] [attn_spec=0] [attn_xflags=2] ...
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned
import torch
args = dict(kv.split("=") for kv in sys.argv[1:])
mfa = Tuned(args.pop("lib", None))
B, H, N, d = map(int, args.pop("shape", "4,32,4096,128").split(","))
data = args.pop("data", "randn")
torch.manual_seed(1)
make = {"randn": torch.randn, "ones": torch.ones}[data]
q, k, v = (make(B, H, N, d, dtype=torch.bfloat16, device="cuda") for _ in range(3))
nwg = B * H * ((N + 255) // 256)
dbg = torch.zeros(nwg * 24, dtype=torch.int64, device="cuda")
mfa.set(attn_dbg=dbg.data_ptr(), **{key: float(val) for key, val in args.items()})
for _ in range(5):
    mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5)
torch.cuda.synchronize()
x = dbg.view(nwg, 2, 12).double().cpu()
T = x[:, 0, 7].median().item()
names = ["M body", "M barrier wait", "X softmax", "X dma wait + prefill", "X barrier wait"]
print(" ".join(sys.argv[1:]), f"(tiles {T:.0f})")
for g in range(2):
    tot = sum(x[:, g, i].median().item() for i in range(5)) / (T - 1)
    print(f"  wave {4*g}: " + "  ".join(f"{n}: {x[:, g, i].median().item() / (T - 1):7.1f}" for i, n in enumerate(names))
          + f"   sum {tot:7.1f}   prologue {x[:, g, 5].median().item():7.0f}  epilogue {x[:, g, 6].median().item():7.0f}")
