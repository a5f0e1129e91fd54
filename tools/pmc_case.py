#!/usr/bin/env python3
"""A few launches of ONE kernel of the library, for a rocprofv3 --pmc pass (tools/pmc_cases.sh).

    python3 tools/pmc_case.py <case> [launches]

cases: softmax32768, softmax8192, cols8192 (softmax over dim 0 of [8192, 8192]: the streamed column
kernel), cols16384, config3 (f16 B8 H16 N1024 d64), wide256 (bf16 B4 H16 N4096 d256), causal
(headline shape under the causal mask), bwd (backward of the headline shape), bwdwide (bf16 B4 H16 N4096 d256 backward), bwdf32 / bwdf32causal (the headline shape's
float32 backward), scale1 (headline shape, unscaled logits).  Prints the algorithmic bytes or flops of one launch.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch  # noqa: E402
import my_flash_attn_cuda as mfa  # noqa: E402


def main():
    case = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    torch.manual_seed(1)
    dev = "cuda"
    if case.startswith("softmax"):
        w = int(case[len("softmax"):])
        x = torch.randn(w, w, device=dev)
        fn = lambda: mfa.my_softmax(x, 1)  # noqa: E731
        note = f"algorithmic_bytes {2 * w * w * 4}"
    elif case.startswith("cols"):
        bf = case.endswith("bf16")
        w = int(case[len("cols"):-4] if bf else case[len("cols"):])
        x = torch.randn(w, w, device=dev)
        if bf:
            x = x.bfloat16()
        fn = lambda: mfa.softmax(x, dim=0)  # noqa: E731
        note = f"algorithmic_bytes {2 * w * w * x.element_size()}"
    elif case in ("config3", "wide256", "causal", "scale1", "headline"):
        B, H, N, d, dt = {"config3": (8, 16, 1024, 64, torch.float16), "wide256": (4, 16, 4096, 256, torch.bfloat16),
                          "causal": (4, 32, 4096, 128, torch.bfloat16), "scale1": (4, 32, 4096, 128, torch.bfloat16),
                          "headline": (4, 32, 4096, 128, torch.bfloat16)}[case]
        q, k, v = (torch.randn(B, H, N, d, dtype=dt, device=dev) for _ in range(3))
        scale = 1.0 if case == "scale1" else d ** -0.5
        fn = lambda: mfa.flash_attn_fwd(q, k, v, scale=scale, causal=case == "causal")  # noqa: E731
        note = f"algorithmic_flops {4 * B * H * N * N * d * (0.5 if case == 'causal' else 1.0):.0f} algorithmic_bytes {4 * B * H * N * d * 2}"
    elif case == "bwd":
        B, H, N, d = 4, 32, 4096, 128
        q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.bfloat16, device=dev) for _ in range(4))
        out, lse = mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5, return_lse=True)
        fn = lambda: mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=d ** -0.5)  # noqa: E731
        note = f"algorithmic_flops {10 * B * H * N * N * d} algorithmic_bytes {8 * B * H * N * d * 2}"
    elif case == "bwdwide":
        B, H, N, d = 4, 16, 4096, 256
        q, k, v, do = (torch.randn(B, H, N, d, dtype=torch.bfloat16, device=dev) for _ in range(4))
        out, lse = mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5, return_lse=True)
        fn = lambda: mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=d ** -0.5)  # noqa: E731
        note = f"algorithmic_flops {10 * B * H * N * N * d} algorithmic_bytes {8 * B * H * N * d * 2}"
    elif case in ("bwdf32", "bwdf32causal"):
        B, H, N, d = 4, 32, 4096, 128
        causal = case.endswith("causal")
        q, k, v, do = (torch.randn(B, H, N, d, device=dev) for _ in range(4))
        out, lse = mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5, return_lse=True, causal=causal)
        fn = lambda: mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=d ** -0.5, causal=causal)  # noqa: E731
        n = min(n, 3)
        note = f"algorithmic_flops {10 * B * H * N * N * d * (0.5 if causal else 1.0):.0f} algorithmic_bytes {8 * B * H * N * d * 4}"
    else:
        raise SystemExit(f"unknown case {case}")
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(case, note, flush=True)


if __name__ == "__main__":
    main()
