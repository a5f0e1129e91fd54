#!/usr/bin/env python3
"""Debug aid: where does the resident-workgroup build (attn_variant=3) differ from variant 2?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned
import torch

t = Tuned()
def run(shape, dtype, causal, wgs, spec):
    B, H, N, d = shape
    torch.manual_seed(3)
    q, k, v = (torch.randn(B, H, N, d).cuda().to(dtype) for _ in range(3))
    t.set(attn_variant=2, attn_spec=spec)
    ref, lref = t.flash_attn_fwd(q, k, v, scale=d ** -0.5, causal=causal, return_lse=True)
    t.set(attn_variant=3, attn_spec=spec, attn_wgs=wgs)
    out, lse = t.flash_attn_fwd(q, k, v, scale=d ** -0.5, causal=causal, return_lse=True)
    torch.cuda.synchronize()
    diff = (out.float() - ref.float()).abs().amax(dim=-1).reshape(B * H, N)   # per head, row
    bad = (diff > 1e-2).nonzero()
    print(f"shape={shape} {dtype} causal={causal} wgs={wgs} spec={spec}: max {float(diff.max()):.3e}  bad rows {len(bad)}"
          f"  lse diff {float((lse - lref).abs().max()):.3e}")
    if len(bad):
        heads = sorted(set(int(b[0]) for b in bad))
        for h in heads[:6]:
            rows = [int(b[1]) for b in bad if int(b[0]) == h]
            print(f"   head {h}: rows {rows[0]}..{rows[-1]} ({len(rows)})")

for spec in (0, 1):
    for causal in (False, True):
        for wgs in (1, 3, 0):
            run((2, 3, 64, 64), torch.float16, causal, wgs, spec)
            run((2, 2, 700, 128), torch.bfloat16, causal, wgs, spec)
            run((1, 3, 1300, 64), torch.float16, causal, wgs, spec)
