#!/usr/bin/env python3
"""The measurements of the reference's benchmark.py (benchmark.py:12-20,35-84), taken the same way
-- 1000 back-to-back calls between two device synchronisations, wall clock, microseconds per call --
on this library: the vendor GEMM entry point, my_softmax with each of the five kernel numbers, and,
beyond what the reference's script runs, my_matmul and my_attention.

    python tools/reference_style_benchmark.py [--n 1024] [--iters 1000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch  # noqa: E402
import my_flash_attn_cuda as mfa  # noqa: E402


def us_per_call(fn, args, iters):
    fn(*args)                                   # (first call: module load, not what is measured)
    torch.cuda.synchronize()
    t0 = time.perf_counter_ns()
    for _ in range(iters):
        fn(*args)
    torch.cuda.synchronize()
    return (time.perf_counter_ns() - t0) / iters / 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=1000)
    a = ap.parse_args()
    n = a.n
    torch.manual_seed(1)
    m1, m2, x = (torch.randn(n, n, device="cuda") for _ in range(3))
    rows = [("my_matmul_cublas", mfa.my_matmul_cublas, (m1, m2)), ("my_matmul", mfa.my_matmul, (m1, m2))]
    rows += [(f"my_softmax kernel_no={kno}", mfa.my_softmax, (x, kno)) for kno in (1, 2, 3, 4, 5)]
    q, v = torch.randn(n, 64, device="cuda"), torch.randn(n, 64, device="cuda")
    k_t = torch.randn(64, n, device="cuda")
    rows.append((f"my_attention ({n} x 64)", mfa.my_attention, (q, k_t, v)))
    print(f"{n} x {n}, {a.iters} calls each")
    for name, fn, args in rows:
        try:
            print(f"{name:28s} {us_per_call(fn, args, a.iters):9.2f} us")
        except RuntimeError as e:                # the reference's script reports a refused kernel number this way
            print(f"{name:28s} {e}")


if __name__ == "__main__":
    main()
