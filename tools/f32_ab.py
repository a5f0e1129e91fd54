#!/usr/bin/env python3
"""A/B of the batched float32 attention forward between two builds of the tuning library:
    python tools/f32_ab.py [other_lib.so]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned
import torch

libs = {"in-tree": Tuned()}
if len(sys.argv) > 1:
    libs[sys.argv[1]] = Tuned(sys.argv[1])
torch.manual_seed(1)
for B, H, N, d in ((4, 32, 4096, 128), (4, 32, 1024, 64), (2, 16, 2048, 256)):
    q, k, v = (torch.randn(B, H, N, d, device="cuda") for _ in range(3))
    res = {n: [] for n in libs}
    for _ in range(5):
        for n, t in libs.items():
            t.set()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t.flash_attn_fwd(q, k, v, scale=d ** -0.5)
            e0.record()
            for _ in range(3):
                t.flash_attn_fwd(q, k, v, scale=d ** -0.5)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 3)
    for n, ts in res.items():
        ms = sorted(ts)[2]
        print(f"{(B, H, N, d)} {n:24s} {ms:8.3f} ms {4 * B * H * N * N * d / ms / 1e9:7.2f} TFLOP/s", flush=True)
