#!/usr/bin/env python3
"""Debug aid: run-to-run determinism of the forced softmax forms on unscaled logits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned
import torch
t = Tuned()
for dt, knobs in ((torch.bfloat16, dict(attn_spec=2)), (torch.float16, dict(attn_spec=2, attn_thr=15)), (torch.float16, dict(attn_spec=2, attn_thr=12)),
                  (torch.float16, dict(attn_spec=1, attn_pre=0, attn_thr=15)), (torch.bfloat16, dict(attn_spec=1, attn_pre=0, attn_thr=60)), (torch.bfloat16, dict())):
    torch.manual_seed(3)
    q, k, v = (torch.randn(4, 32, 4096, 128, dtype=dt, device="cuda") for _ in range(3))
    t.set(attn_variant=3, **knobs)
    outs = [t.flash_attn_fwd(q, k, v, scale=1.0) for _ in range(6)]
    torch.cuda.synchronize()
    bad = [int((~torch.isfinite(o.float())).any(dim=-1).sum()) for o in outs]
    diff = [int((outs[0] != o).sum()) for o in outs[1:]]
    print(dt, knobs, "nonfinite rows per run", bad, "elements differing from run 0", diff)
