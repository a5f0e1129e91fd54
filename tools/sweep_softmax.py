#!/usr/bin/env python3
"""Sweeps launch configurations of the register-resident softmax kernel (GPU box).

MYFA_SOFTMAX_SHAPE = "threads_per_row,items_per_thread,map,nt,pad" (see softmax_rows.hip).
Prints the best few per size and the library's default choice.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
import my_flash_attn_cuda as mfa

SHAPES = [(256, 4), (256, 8), (512, 2), (512, 4), (512, 8), (1024, 2), (1024, 4), (1024, 8)]


def timeit(xs, iters=20):
    for i in range(3):
        mfa.my_softmax(xs[i % len(xs)], 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(ev):
        a.record(); mfa.my_softmax(xs[i % len(xs)], 1); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


for n in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768]:
    nbuf = 3 if n <= 8192 else 1
    xs = [torch.randn(n, n, device="cuda") for _ in range(nbuf)]
    gbs = lambda ms: 2 * n * n * 4 / (ms * 1e-3) / 1e9
    os.environ.pop("MYFA_SOFTMAX_SHAPE", None)
    default = gbs(timeit(xs))
    res = []
    for tpr, v in SHAPES:
        if tpr * v * 4 < n or tpr * v * 4 > 4 * n:
            continue
        for mp in (0, 1):
            for nt in (2, 3):
                for pad in (0, 27000, 54000):
                    os.environ["MYFA_SOFTMAX_SHAPE"] = f"{tpr},{v},{mp},{nt},{pad}"
                    res.append((gbs(timeit(xs)), os.environ["MYFA_SOFTMAX_SHAPE"]))
    os.environ.pop("MYFA_SOFTMAX_SHAPE", None)
    res.sort(reverse=True)
    print(f"n={n}: default {default:.0f} | " + "  ".join(f"[{k}] {g:.0f}" for g, k in res[:6]) + "  GB/s", flush=True)
    del xs
    torch.cuda.empty_cache()
