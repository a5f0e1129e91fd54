#!/usr/bin/env python3
"""my_attention (float32, one head) with the keys split over workgroups: device time per call
(HIP events around 20 back-to-back calls) for several workgroup-count targets."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned
import torch

t = Tuned()
torch.manual_seed(1)
for n, d in ((128, 64), (512, 64), (1024, 64), (1025, 65), (1025, 99), (2048, 128)):
    q, k, v = (torch.randn(n, d, device="cuda") for _ in range(3))
    kt = k.t().contiguous()
    for label, kn, ws in (("no workspace", {}, False), ("256", {"attn_f32_wgs": 256}, True), ("320", {"attn_f32_wgs": 320}, True), ("384", {"attn_f32_wgs": 384}, True), ("512", {"attn_f32_wgs": 512}, True),
                          ("1024", {"attn_f32_wgs": 1024}, True), ("2048", {"attn_f32_wgs": 2048}, True)):
        t.set(**kn)
        for _ in range(5):
            t.my_attention(q, kt, v, ws)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                t.my_attention(q, kt, v, ws)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"N={n} d={d} target {label:>12s}: {sorted(ts)[3]:7.2f} us per call (back to back)", flush=True)
