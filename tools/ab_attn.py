#!/usr/bin/env python3
"""Interleaved A/B of attention-kernel variants in ONE process (GPU box).

    python tools/ab_attn.py "attn_variant=1" "attn_variant=2" "attn_variant=2,attn_thr=0" [--shape B,H,N,d] [--dtype bf16] [--rounds 7]

Each spec is a comma list of tuning knobs (csrc/tuning.h) set on the tuning flavour of the
library around its launches.  Prints median / min ms and TFLOP/s per spec and the max
|difference| of each spec's output from the first spec's.  --data ones|zeros times the same
launches on constant operands (the chip holds a higher clock there).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tuned import Tuned  # noqa: E402
import torch  # noqa: E402


def knobs(spec):
    return {k: float(v) for k, v in (kv.split("=") for kv in filter(None, spec.split(","))) if k != "lib"}


_LIBS = {}


def lib_of(spec):
    """`lib=path` in a spec loads another build of the tuning flavour (default: the in-tree one)."""
    path = dict(kv.split("=") for kv in filter(None, spec.split(","))).get("lib")
    if path not in _LIBS:
        _LIBS[path] = Tuned(path)
    return _LIBS[path]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--shape", default="4,32,4096,128")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--scale", type=float, default=None)
    ap.add_argument("--data", default="randn", choices=["randn", "ones", "zeros"])
    ap.add_argument("--causal", action="store_true", help="causal mask; TFLOP/s then count half the flops")
    a = ap.parse_args()
    B, H, N, d = map(int, a.shape.split(","))
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    scale = a.scale if a.scale is not None else d ** -0.5
    torch.manual_seed(1)
    make = {"randn": torch.randn, "ones": torch.ones, "zeros": torch.zeros}[a.data]
    q, k, v = (make(B, H, N, d, dtype=dt, device="cuda") for _ in range(3))
    flops = 4.0 * B * H * N * N * d * (0.5 if a.causal else 1.0)
    outs, times = {}, {s: [] for s in a.specs}
    for s in a.specs:
        t = lib_of(s)
        t.set(**knobs(s))
        outs[s] = t.flash_attn_fwd(q, k, v, scale=scale, causal=a.causal).float()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for s in a.specs:
            t = lib_of(s)
            t.set(**knobs(s))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                t.flash_attn_fwd(q, k, v, scale=scale, causal=a.causal)
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / a.iters)
    base = outs[a.specs[0]]
    for s in a.specs:
        ts = sorted(times[s])
        med, mn = ts[len(ts) // 2], ts[0]
        print(f"{s:34s} median {med:8.4f} ms  {flops / med / 1e9:8.1f} TF   min {mn:8.4f} ms  {flops / mn / 1e9:8.1f} TF"
              f"   maxdiff_vs_first {float((outs[s] - base).abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    main()
