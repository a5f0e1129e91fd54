#!/usr/bin/env python3
"""Interleaved A/B of attention-kernel variants in ONE process (GPU box).

    python tools/ab_attn.py "VARIANT=0" "VARIANT=1" "VARIANT=1,THR=8" [--shape B,H,N,d] [--dtype bf16] [--rounds 7]

Each spec sets MYFA_ATTN_<KEY>=<value> around its launches.  Prints median / min ms and
TFLOP/s per spec and the max |difference| of each spec's output from the first spec's.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch  # noqa: E402
import my_flash_attn_cuda as mfa  # noqa: E402


def set_env(spec):
    for k in list(os.environ):
        if k.startswith("MYFA_ATTN_"):
            del os.environ[k]
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        os.environ["MYFA_ATTN_" + k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--shape", default="4,32,4096,128")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--scale", type=float, default=None)
    ap.add_argument("--causal", action="store_true", help="causal mask; TFLOP/s then count half the flops")
    a = ap.parse_args()
    B, H, N, d = map(int, a.shape.split(","))
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    scale = a.scale if a.scale is not None else d ** -0.5
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=dt, device="cuda") for _ in range(3))
    flops = 4.0 * B * H * N * N * d * (0.5 if a.causal else 1.0)
    outs, times = {}, {s: [] for s in a.specs}
    for s in a.specs:
        set_env(s)
        outs[s] = mfa.flash_attn_fwd(q, k, v, scale=scale, causal=a.causal).float()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for s in a.specs:
            set_env(s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                mfa.flash_attn_fwd(q, k, v, scale=scale, causal=a.causal)
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / a.iters)
    base = outs[a.specs[0]]
    for s in a.specs:
        ts = sorted(times[s])
        med, mn = ts[len(ts) // 2], ts[0]
        print(f"{s:28s} median {med:8.4f} ms  {flops / med / 1e9:8.1f} TF   min {mn:8.4f} ms  {flops / mn / 1e9:8.1f} TF"
              f"   maxdiff_vs_first {float((outs[s] - base).abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    main()
