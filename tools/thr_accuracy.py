#!/usr/bin/env python3
"""Accuracy of the bf16/f16 forward against float64 attention on the rounded inputs, as a function of
the deferred-max limit (GPU box, tuning flavour): max |error| of O over several seeds, at the
reference's unscaled logits (scale = 1.0, tests/test_flash_attn.py:165) and at d^-0.5.

    python tools/thr_accuracy.py [lib=path] [dtype=bf16] [spec=1] [thrs=0,8,12,16,20,30,45] [seeds=6]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned
import torch
args = dict(kv.split("=") for kv in sys.argv[1:])
mfa = Tuned(args.pop("lib", None))
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.pop("dtype", "bf16")]
spec = int(args.pop("spec", "1"))
thrs = [float(x) for x in args.pop("thrs", "0,8,12,16,20,30,45").split(",")]
seeds = int(args.pop("seeds", "6"))
B, H, N, d = 1, 8, 2048, 128
for scale in (1.0, d ** -0.5):
    worst = {t: 0.0 for t in thrs}
    over = {t: 0 for t in thrs}
    for seed in range(seeds):
        torch.manual_seed(100 + seed)
        q, k, v = (torch.randn(B, H, N, d, device="cuda").to(dt) for _ in range(3))
        ref = torch.softmax((q.double() @ k.double().transpose(-1, -2)) * scale, dim=-1) @ v.double()
        for t in thrs:
            mfa.set(attn_spec=spec, attn_thr=t)
            out = mfa.flash_attn_fwd(q, k, v, scale=scale).double()
            err = (out - ref).abs()
            worst[t] = max(worst[t], float(err.max()))
            over[t] += int((err > 2e-2).sum())
    print(f"scale={scale:.4f} spec={spec} dtype={dt}: " + "  ".join(f"thr {t:g}: max {worst[t]:.4f} (>2e-2: {over[t]})" for t in thrs), flush=True)
