#!/usr/bin/env python3
"""Benchmark of the attention-forward / row-softmax hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|config3|config5]

N > 1 runs one rank per GPU under torch.distributed.run (RCCL).  Started plainly
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) the script starts that launcher
itself as a child process -- before anything touches the GPU -- relays rank 0's JSON line and
exits with the child's status; started by a launcher it is a rank and checks that the world it
finds is the --gpus it was asked for.  Every rank runs the same per-GPU workload on its own
(batch x head) shard -- weak scaling, no collective on the timed path.  Rank 0 prints ONE JSON line:

  metric / value   whole-job attention-forward TFLOP/s (4*B*H*N^2*d flops per pass per GPU),
                   inputs resident in HBM, max-over-ranks wall time between barriers
  roofline         the attention kernel against the dense bf16 MFMA peak (2.5 PFLOP/s):
                   algorithmic flops per launch / mean launch duration = one HIP event pair
                   around the K timed launches, on the launch stream, / K
  cpu_baseline     the C oracle (float32 arithmetic, OpenMP) timed on this host's cores on
                   a bounded sample of the same workload (rank 0, N=1 only)
  softmax_sweep    row-softmax GB/s (algorithmic 8 B/element) at n x n, n = 4096..32768,
                   against 8 TB/s HBM peak (rank 0, N=1 only)
  softmax_1024     my_softmax at 1024 x 1024 timed the way the reference's benchmark.py does, beside the
                   reference's published T4 numbers (the only timings it publishes)
  softmax_dim      softmax along a non-last dimension (8 algorithmic B/element in float32, 4 in bfloat16), and
  matmul_f32       my_matmul at n^3 against the 157.3 TFLOP/s float32 MFMA peak beside the vendor GEMM
  cpu_config1      torch CPU timings of the CPU reference configuration (softmax 1024^2, naive attention
                   N=128 d=64), for reference beside the GPU numbers
  attention_f32    the float32 path (float64-MFMA logits + float32-MFMA P.V): the reference's
                   my_attention(Q, K_t, V) at N=128 d=64 (config 1) and N=1024 d=64 (its largest
                   test shape), and the batched float32 flavour of the headline shape, against
                   104.8 TFLOP/s = 4 / (2/78.6 + 2/157.3), the two matrix rates in series
  attention_backward  the backward (delta, dQ, dK/dV kernels) of the same shape: ms, TFLOP/s over 2.5x
                   the forward's flops, and PyTorch's SDPA backward on the same tensors
  attention_f32_backward  the float32 backward of the headline shape (float32 matrix instructions: 157.3
                   TFLOP/s peak; five products counted, seven executed) beside PyTorch's float32 SDPA backward
  attention_causal the same shape under the causal mask, forward and backward (half the products count)
  config5_strong_scaling  configs[4] as BASELINE states it: ONE bf16 B=8 H=32 N=16384 d=128 problem whose
                   256 (batch, head) problems are split over the ranks (sharding.head_range), aggregate
                   TFLOP/s = 35.18 TFLOP / max-over-ranks time per pass -- emitted for every --gpus N
  config3 / headline_scale_one  configs[2] (f16 B8 H16 N1024 d64) and the headline shape with the
                   reference's unscaled logits (scale = 1.0), N = 1 only
  wide_head_dim    bf16 B4 H16 N4096 d256 (the 256-column build for head dims above 128), N = 1 only;
                   wide_head_dim_backward: its backward (2.5x the forward's algorithmic flops) beside
                   PyTorch's scaled_dot_product_attention backward
  power_context    the same attention launch on all-ones operands, and the vendor bf16 GEMM
                   (torch.matmul -> hipBLASLt, 8192^3) on random and all-ones operands: the chip
                   lowers its clock under the power limit when operands are random, so these
                   show how much of the distance to the 2.5 PFLOP/s peak is schedule and how
                   much is the power limit any dense MFMA kernel meets on this data
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "flash-attention_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/f16 MFMA peak, MI355X
HBM_PEAK_GBS = 8000.0       # HBM3E peak
PRIMING = 100               # untimed launches before the warm-up steps (the clocks settle over ~55 launches, ~50 ms)

WORKLOADS = {
    # name: (B, H, N, d, dtype) -- per GPU
    "headline": (4, 32, 4096, 128, torch.bfloat16),    # the configuration the headline metric is quoted on
    "config3": (8, 16, 1024, 64, torch.float16),       # configs[2]
    "config5": (1, 32, 16384, 128, torch.bfloat16),    # configs[4]: the shard one of 8 GPUs gets
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def attn_flops(B, H, N, d, causal=False):
    """Algorithmic flops; under the causal mask the visible half of the N x N products."""
    return 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)


def bench_attention(mfa, B, H, N, d, dtype, steps, warmup, world, scale, causal=False, use_dist=False):
    torch.manual_seed(1 + int(os.environ.get("RANK", "0")))
    q, k, v = (torch.randn(B, H, N, d, dtype=dtype, device="cuda") for _ in range(3))
    # Untimed: PRIMING launches bring the clocks and caches to their running state whatever W is,
    # then the W warm-up steps of the contract.  From idle the kernel takes 0.90, 1.06, 1.21 ms for
    # its first launches and then comes down slowly -- 0.92 ms at launch 20, 0.89 at 30, 0.87 from
    # launch ~55 on, where it stays (kernel trace of 223 launches, profiles/) -- so fewer priming
    # launches would time the power management's transient, not the kernel.
    for _ in range(PRIMING):
        out = mfa.flash_attn_fwd(q, k, v, scale=scale, causal=causal)
    for _ in range(warmup):
        out = mfa.flash_attn_fwd(q, k, v, scale=scale, causal=causal)
    # One pair of HIP events around the timed region, on the stream the launches go to: an event
    # between two launches is a barrier packet that drains the GPU before the next kernel starts, so
    # a pair per launch would put a bubble into every step.  Mean launch duration = region / K (it
    # includes the gaps between back-to-back launches, so it bounds the kernel's time from above).
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    start.record()
    for i in range(steps):
        out = mfa.flash_attn_fwd(q, k, v, scale=scale, causal=causal)
    stop.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    mean_ms = start.elapsed_time(stop) / steps
    assert bool(torch.isfinite(out.float()).all())
    return wall, mean_ms, (q, k, v, out)


def bench_config5_strong(mfa, rank, world, use_dist, steps=4, warm=2):
    """configs[4]: the fixed bf16 B=8 H=32 N=16384 d=128 problem (35.18 TFLOP per pass), its 256
    (batch, head) problems split contiguously over the ranks; no collective on the timed path."""
    from my_flash_attn_cuda import sharding
    B, H, N, d = 8, 32, 16384, 128
    lo, hi = sharding.head_range(B * H, rank, world)
    torch.manual_seed(50 + rank)
    q, k, v = (torch.randn(1, hi - lo, N, d, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    scale = d ** -0.5
    for _ in range(warm):
        out = mfa.flash_attn_fwd(q, k, v, scale=scale)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = mfa.flash_attn_fwd(q, k, v, scale=scale)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    assert bool(torch.isfinite(out[0, 0].float()).all())
    heads_by_rank = [hi - lo]
    if use_dist:
        hb = torch.zeros(world, dtype=torch.int32, device="cuda")
        hb[rank] = hi - lo
        dist.all_reduce(hb)
        heads_by_rank = hb.tolist()
    assert sum(heads_by_rank) == B * H
    total = attn_flops(B, H, N, d)
    tf = total * steps / wall / 1e12
    del q, k, v, out
    torch.cuda.empty_cache()
    return {"workload": f"bf16 B={B} H={H} N={N} d={d}, one problem split by (batch x head) over {world} rank(s)",
            "scaling": "strong", "heads_total": B * H, "heads_this_rank": hi - lo, "heads_by_rank": heads_by_rank, "steps": steps,
            "ms_per_pass": round(wall / steps * 1e3, 3), "TFLOP/s": round(tf, 1),
            "frac_mfma_peak_per_gpu": round(tf / world / MFMA_PEAK_TFLOPS, 4), "flops_per_pass": total}


def bench_shape(mfa, B, H, N, d, dtype, scale, iters=30):
    """One more attention-forward shape, device time per launch (median of HIP-event brackets)."""
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, H, N, d, dtype=dtype, device="cuda") for _ in range(3))
    ms = _median_ms(lambda: mfa.flash_attn_fwd(q, k, v, scale=scale), iters=iters, warm=PRIMING)
    tf = attn_flops(B, H, N, d) / ms / 1e9
    del q, k, v
    torch.cuda.empty_cache()
    return {"shape": [B, H, N, d], "dtype": {torch.bfloat16: "bf16", torch.float16: "f16"}[dtype], "scale": scale,
            "us": round(ms * 1e3, 2), "TFLOP/s": round(tf, 1), "frac_mfma_peak": round(tf / MFMA_PEAK_TFLOPS, 4)}


def profiled_traffic():
    """HBM bytes per launch of the headline kernel from the newest committed counter file
    (profiles/rNN_attn_headline_pmc.txt, written by tools/pmc_attn.sh on a GPU box): reads =
    TCC_EA0_RDREQ x 128 B (on gfx950 a wide streaming read request is tallied as 64 B by the
    profiler's FETCH_SIZE, half its size), writes = 64 B per 64-byte write request + 32 B per other
    one.  None when no such file has both counters."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_attn_headline_pmc.txt")))
    for f in reversed(files):
        vals = {}
        with open(f) as fh:
            for line in fh:
                m = re.match(r"^(TCC_EA0_\w+)\s+([0-9.]+)", line)
                if m:
                    vals[m.group(1)] = float(m.group(2))
        if "TCC_EA0_RDREQ_sum" in vals and "TCC_EA0_WRREQ_sum" in vals:
            rd = vals["TCC_EA0_RDREQ_sum"] * 128.0
            w64 = vals.get("TCC_EA0_WRREQ_64B_sum", vals["TCC_EA0_WRREQ_sum"])
            wr = w64 * 64.0 + (vals["TCC_EA0_WRREQ_sum"] - w64) * 32.0
            return {"bytes": rd + wr, "read_bytes": rd, "write_bytes": wr, "source": os.path.relpath(f, ROOT)}
    return None


def profiled_softmax_traffic():
    """HBM bytes per launch of the row-softmax kernel at n = 8192 and 32768 from the newest committed
    counter file of the other kernels (profiles/rNN_other_kernels_pmc.txt, written by
    tools/pmc_round3.sh on a GPU box): sections `=== softmax<n> ===`, reads = TCC_EA0_RDREQ x 128 B,
    writes = TCC_EA0_WRREQ x 64 B (all of them 64-byte requests there).  {} when there is no such file."""
    import glob
    import re
    out = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_other_kernels_pmc.txt")))
    if not files:
        return out
    n, vals = None, {}

    def close():
        if n is not None and "TCC_EA0_RDREQ_sum" in vals and "TCC_EA0_WRREQ_sum" in vals:
            out[n] = {"bytes": vals["TCC_EA0_RDREQ_sum"] * 128.0 + vals["TCC_EA0_WRREQ_sum"] * 64.0,
                      "source": os.path.relpath(files[-1], ROOT)}
    with open(files[-1]) as fh:
        for line in fh:
            m = re.match(r"^=== (\w+)", line)
            if m:
                close()
                sm = re.match(r"^softmax(\d+)$", m.group(1))
                n, vals = (int(sm.group(1)) if sm else None), {}
                continue
            m = re.match(r"^(TCC_EA0_\w+)\s+([0-9.]+)", line)
            if m and n is not None:
                vals[m.group(1)] = float(m.group(2))
    close()
    return out


def bench_softmax_sweep(mfa, iters=20):
    res = []
    traffic = profiled_softmax_traffic()
    for n in (4096, 8192, 16384, 32768):
        nbuf = 3 if n <= 8192 else 1      # rotate inputs larger than the 256 MiB Infinity Cache
        xs = [torch.randn(n, n, dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        for i in range(3):
            out = mfa.my_softmax(xs[i % nbuf], 1)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(ev):
            a.record()
            out = mfa.my_softmax(xs[i % nbuf], 1)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        med = ms[len(ms) // 2]
        gbs = 2.0 * n * n * 4 / (med * 1e-3) / 1e9
        res.append({"n": n, "us": round(med * 1e3, 2), "GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                    # HBM bytes per launch from the committed counter passes of this size (None where none
                    # was taken), beside the algorithmic 8 B/element
                    "traffic": traffic[n]["bytes"] if n in traffic else None, "algorithmic_bytes": 2.0 * n * n * 4,
                    "traffic_source": traffic[n]["source"] if n in traffic else None})
        del xs, out
        torch.cuda.empty_cache()
    return res


def bench_softmax_dim(mfa):
    """softmax along a non-last dimension: register-resident column blocks (256 threads; 1024 for the
    2048-row columns), streamed columns, and tall narrow tensors (rows of a column block split over
    workgroups)."""
    res = []
    f32, bf16 = torch.float32, torch.bfloat16
    for shape, dim, dt in (((256, 64, 64, 64), 1, f32), ((64, 128, 16384), 1, f32), ((4, 8, 2048, 2048), 2, f32), ((8192, 8192), 0, f32),
                           ((16384, 4096), 0, f32), ((100000, 512), 0, f32), ((4, 8, 2048, 2048), 2, bf16), ((8192, 8192), 0, bf16)):
        xs = [torch.randn(*shape, dtype=torch.float32, device="cuda").to(dt) for _ in range(3)]
        i = [0]

        def run():
            i[0] += 1
            return mfa.softmax(xs[i[0] % 3], dim=dim)
        ms = _median_ms(run, iters=20)
        gbs = 2.0 * xs[0].numel() * xs[0].element_size() / (ms * 1e-3) / 1e9
        res.append({"shape": list(shape), "dim": dim, "dtype": "f32" if dt == f32 else "bf16", "us": round(ms * 1e3, 2), "GB/s": round(gbs, 1),
                    "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})
        del xs
        torch.cuda.empty_cache()
    return res


def bench_matmul_f32(mfa):
    """my_matmul (the library's float32 MFMA kernels) against the 157.3 TFLOP/s float32 MFMA peak,
    beside the vendor GEMM on the same operands."""
    res = []
    for n in (1024, 4096, 8192):
        a, b = torch.randn(n, n, device="cuda"), torch.randn(n, n, device="cuda")
        ms = _median_ms(lambda: mfa.my_matmul(a, b), iters=10)
        ms_v = _median_ms(lambda: torch.matmul(a, b), iters=10)
        tf = 2.0 * n ** 3 / (ms * 1e-3) / 1e12
        res.append({"n": n, "us": round(ms * 1e3, 1), "TFLOP/s": round(tf, 1), "frac_mfma_f32_peak": round(tf / 157.3, 4),
                    "vendor_gemm_TFLOP/s": round(2.0 * n ** 3 / (ms_v * 1e-3) / 1e12, 1)})
        del a, b
    torch.cuda.empty_cache()
    return res


def _median_ms(fn, iters=15, warm=3, settle_s=0.06):
    """Median HIP-event time of fn().  After the `warm` calls the GPU is kept busy with fn for
    another settle_s seconds (at most 200 calls) before anything is timed: coming from a different
    kind of load (or from idle) the clocks take ~50 ms to settle, and a 4096^3 SGEMM timed over its
    first 13 launches reads 125 TFLOP/s where every later launch runs at 141."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-6)
    for _ in range(min(200, int(settle_s / one))):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def power_context(mfa, B, H, N, d, dtype, scale):
    """Untimed context: same kernel on all-ones data; vendor GEMM on random / all-ones data."""
    ones = torch.ones(B, H, N, d, dtype=dtype, device="cuda")
    ms = _median_ms(lambda: mfa.flash_attn_fwd(ones, ones, ones, scale=scale))
    res = {"attn_all_ones_TFLOP/s": round(attn_flops(B, H, N, d) / ms / 1e9, 1)}
    del ones
    n = 8192
    for name, make in (("randn", lambda: torch.randn(n, n, device="cuda")), ("ones", lambda: torch.ones(n, n, device="cuda"))):
        a, b = make().to(dtype), make().to(dtype)
        ms = _median_ms(lambda: torch.matmul(a, b))
        res[f"vendor_gemm_{n}_{name}_TFLOP/s"] = round(2.0 * n ** 3 / ms / 1e9, 1)
        del a, b
    torch.cuda.empty_cache()
    return res


F32_PEAK_TFLOPS = 4.0 / (2.0 / 78.6 + 2.0 / 157.3)   # f64 MFMA for Q.K^T and f32 MFMA for P.V, in series


def bench_attention_f32(mfa):
    res = []
    torch.manual_seed(1)
    for n, d in ((128, 64), (1024, 64)):
        q, k, v = (torch.randn(n, d, device="cuda") for _ in range(3))
        kt = k.t().contiguous()
        ms = _median_ms(lambda: mfa.my_attention(q, kt, v), iters=30)
        tf = 4.0 * n * n * d / ms / 1e9
        # ... and the way the reference times its kernels (benchmark.py:12-20): calls back to back
        # between two synchronisations, mean per call with launch overhead and the output allocation
        calls = 1000
        torch.cuda.synchronize()
        t0 = time.perf_counter_ns()
        for _ in range(calls):
            mfa.my_attention(q, kt, v)
        torch.cuda.synchronize()
        b2b = (time.perf_counter_ns() - t0) / calls / 1e3
        res.append({"call": f"my_attention N={n} d={d}", "us": round(ms * 1e3, 2), "us_per_call_back_to_back": round(b2b, 2),
                    "TFLOP/s": round(tf, 3), "frac_mfma_f64_f32_peak": round(tf / F32_PEAK_TFLOPS, 4)})
    B, H, N, d = 4, 32, 4096, 128
    q, k, v = (torch.randn(B, H, N, d, device="cuda") for _ in range(3))
    ms = _median_ms(lambda: mfa.flash_attn_fwd(q, k, v, scale=d ** -0.5), iters=5, warm=2)
    tf = attn_flops(B, H, N, d) / ms / 1e9
    res.append({"call": f"flash_attn_fwd float32 B={B} H={H} N={N} d={d}", "us": round(ms * 1e3, 1), "TFLOP/s": round(tf, 2),
                "frac_mfma_f64_f32_peak": round(tf / F32_PEAK_TFLOPS, 4)})
    del q, k, v
    torch.cuda.empty_cache()
    return res


def bench_softmax_1024(mfa, calls=1000):
    """The one timing the reference publishes (README.md:44-54, taken by benchmark.py:12-20):
    1024 x 1024 float32, `calls` back-to-back my_softmax calls between two synchronisations, mean
    microseconds per call including launch overhead and the output allocation."""
    torch.manual_seed(1)
    x = torch.randn(1024, 1024, dtype=torch.float32, device="cuda")
    for _ in range(20):
        mfa.my_softmax(x, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter_ns()
    for _ in range(calls):
        mfa.my_softmax(x, 5)
    torch.cuda.synchronize()
    us = (time.perf_counter_ns() - t0) / calls / 1e3
    dev_us = _median_ms(lambda: mfa.my_softmax(x, 5), iters=50) * 1e3
    return {"method": f"{calls} back-to-back calls between synchronisations (the reference's benchmark.py)",
            "us_per_call": round(us, 2), "us_per_call_single_event_bracketed": round(dev_us, 2),
            "reference_published_us": {"kernel_1": 110, "kernel_5": 71, "hardware": "NVIDIA T4"}}


def bench_backward(mfa, B, H, N, d, dtype, scale):
    """Forward + backward of the headline shape (untimed context of the headline number): the
    backward's three kernels against 2.5x the forward's algorithmic flops, and PyTorch's own
    scaled_dot_product_attention backward on the same tensors."""
    torch.manual_seed(2)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=dtype, device="cuda") for _ in range(4))
    out, lse = mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True)
    ms = _median_ms(lambda: mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=scale), iters=10)
    res = {"ms": round(ms, 3), "TFLOP/s_algorithmic": round(2.5 * attn_flops(B, H, N, d) / ms / 1e9, 1)}
    qf, kf, vf = (x.detach().clone().requires_grad_(True) for x in (q, k, v))
    o = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, scale=scale)
    ms_t = _median_ms(lambda: torch.autograd.grad(o, (qf, kf, vf), do, retain_graph=True), iters=5, warm=2)
    res["torch_sdpa_backward_ms"] = round(ms_t, 3)
    del q, k, v, do, out, lse, qf, kf, vf, o
    torch.cuda.empty_cache()
    return res


def bench_causal(mfa, B, H, N, d, dtype, scale):
    """The headline shape under the causal mask, forward and backward, counting the visible half of
    the products (untimed context of the headline number)."""
    torch.manual_seed(3)
    q, k, v, do = (torch.randn(B, H, N, d, dtype=dtype, device="cuda") for _ in range(4))
    out, lse = mfa.flash_attn_fwd(q, k, v, scale=scale, return_lse=True, causal=True)
    ms_f = _median_ms(lambda: mfa.flash_attn_fwd(q, k, v, scale=scale, causal=True), iters=30, warm=PRIMING)
    ms_b = _median_ms(lambda: mfa.flash_attn_bwd(do, q, k, v, out, lse, scale=scale, causal=True), iters=15, warm=5)
    half = 0.5 * attn_flops(B, H, N, d)
    res = {"forward_ms": round(ms_f, 3), "forward_TFLOP/s": round(half / ms_f / 1e9, 1),
           "backward_ms": round(ms_b, 3), "backward_TFLOP/s_algorithmic": round(2.5 * half / ms_b / 1e9, 1)}
    del q, k, v, do, out, lse
    torch.cuda.empty_cache()
    return res


def cpu_baseline(B, H, N, d, scale, budget_s=12.0):
    """C oracle, float32 arithmetic, OpenMP: heads of the same workload until ~budget_s."""
    import numpy as np
    from oracle import c_oracle
    c_oracle.build()
    cores = min(16, os.cpu_count() or 1, c_oracle.max_threads())
    c_oracle.set_threads(cores)
    rng = np.random.default_rng(1)
    q, k, v = (rng.standard_normal((1, N, d), dtype=np.float32) for _ in range(3))
    t0 = time.perf_counter()
    c_oracle.attention(q, k, v, scale=scale, f32_arith=True)
    one = time.perf_counter() - t0
    heads = int(max(1, min(B * H, budget_s / max(one, 1e-3))))
    q, k, v = (rng.standard_normal((heads, N, d), dtype=np.float32) for _ in range(3))
    t0 = time.perf_counter()
    c_oracle.attention(q, k, v, scale=scale, f32_arith=True)
    dt = time.perf_counter() - t0
    return {"value": round(heads * attn_flops(1, 1, N, d) / dt / 1e12, 5), "unit": "TFLOP/s", "cores": cores,
            "kind": "port", "seconds": round(dt, 2),
            "sample": f"{heads} of the {B * H} (batch, head) problems of the workload, float32, N={N} d={d}"}


def cpu_config1():
    """The CPU reference configuration (softmax 1024^2, naive attention N=128 d=64) on the host
    with torch CPU ops (not part of any GPU number)."""
    torch.manual_seed(1)
    x = torch.randn(1024, 1024)
    q, k, v = (torch.randn(128, 64) for _ in range(3))

    def med(f, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[n // 2]
    t_s = med(lambda: torch.softmax(x, 1), 50)
    t_a = med(lambda: torch.softmax(q @ k.t(), 1) @ v, 200)
    return {"threads": torch.get_num_threads(), "cpu_count": os.cpu_count(),
            "softmax_1024x1024_us": round(t_s * 1e6, 1), "softmax_GB/s": round(8 * 1024 * 1024 / t_s / 1e9, 2),
            "naive_attention_N128_d64_us": round(t_a * 1e6, 1),
            "naive_attention_GFLOP/s": round(4 * 128 * 128 * 64 / t_a / 1e9, 2)}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run with N ranks of
    this script as a CHILD process (this process has not touched the GPU and never will), pass
    rank 0's one JSON line through and return the child's exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    log(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            log(ln)
    if proc.returncode != 0:
        log(f"[bench] the launcher exited with status {proc.returncode}")
        return proc.returncode
    if len(lines) != 1:
        log(f"[bench] expected one JSON line from rank 0, got {len(lines)}")
        return 1
    got = json.loads(lines[0]).get("n_gpus")
    if got != n:
        log(f"[bench] the ranks report n_gpus={got}, asked for {n}")
        return 1
    print(lines[0], flush=True)
    return 0


def stub_rank(args, world, rank):
    """--stub: the launcher / rendezvous / one-line-JSON plumbing with no GPU in it (gloo, CPU
    tensors, a token workload), so that the multi-rank path of this script is testable on a box
    without GPUs.  Measures nothing: metric and data say so."""
    dist.init_process_group("gloo")
    ones = torch.ones(1)
    dist.all_reduce(ones)
    heads = torch.zeros(world, dtype=torch.int64)
    from my_flash_attn_cuda import sharding   # loads the library (no GPU call), as every rank does
    lo, hi = sharding.head_range(256, rank, world)
    heads[rank] = hi - lo
    dist.all_reduce(heads)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        torch.ones(64, 64) @ torch.ones(64, 64)
    dist.barrier()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "stub (launcher self-test, nothing measured)", "value": 0.0, "unit": "none",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(wall.item()) / max(args.steps, 1) * 1e3, 4), "data": "stub",
                          "verify": {"ranks": int(ones.item()), "heads_by_rank": heads.tolist()}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", default="rsqrt_d", choices=["rsqrt_d", "one"],
                    help="logit scale: d^-0.5 (usual) or 1.0 (the reference's unscaled semantics)")
    ap.add_argument("--causal", action="store_true", help="causal mask (not the headline metric; counts half the flops)")
    ap.add_argument("--no-extras", action="store_true", help="headline launches only: skip config 5, the softmax sweep and the CPU baselines")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the barriers / reductions / checksum gather "
                         "even with one rank (rehearsal of the multi-GPU path on a one-GPU box)")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test: gloo, CPU tensors, a token workload; measures nothing")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_dist:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report "
            f"a {args.gpus}-GPU number measured on {world}")
        sys.exit(2)
    if args.stub:
        return stub_rank(args, world, rank)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    # Rank 0's stdout carries exactly one line, the JSON.  RCCL prints a version banner to the
    # process's stdout when the first communicator is built, so file descriptor 1 points at stderr
    # until the JSON line is due.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # The first collective builds the communicator (hundreds of ms).  Done here, untimed, so that
        # the barrier in front of the timed steps is a short one: a GPU left idle for that long
        # drops its clocks, and the first ~10 ms of launches after it would run slow.
        dist.barrier()
        torch.cuda.synchronize()

    import my_flash_attn_cuda as mfa   # raises if the HIP library is not built

    B, H, N, d, dtype = WORKLOADS[args.workload]
    scale = d ** -0.5 if args.scale == "rsqrt_d" else 1.0
    wall, mean_ms, (_, _, _, out) = bench_attention(mfa, B, H, N, d, dtype, args.steps, args.warmup, world, scale, args.causal, use_dist)
    # Verification only, outside the timed region: every rank's per-head output checksums are
    # gathered over RCCL (xGMI); all ranks must end up with the same, finite table.
    from my_flash_attn_cuda import sharding
    torch.cuda.synchronize()
    tg = time.perf_counter()
    table = sharding.gather_head_checksums(out.reshape(1, B * H, N, d), world * B * H, rank, world, force=use_dist)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - tg) * 1e3
    assert bool(torch.isfinite(table).all()) and table.shape == (world * B * H, 2)
    # what RCCL saw: an all-reduce of 1 over the ranks
    ranks_seen = 1
    if use_dist:
        one = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        assert ranks_seen == world, f"all-reduce of 1 over the ranks gave {ranks_seen}, world is {world}"
    flops = attn_flops(B, H, N, d, args.causal)
    value = world * args.steps * flops / wall / 1e12
    achieved = flops / (mean_ms * 1e-3) / 1e12
    del out
    torch.cuda.empty_cache()
    # (--no-extras leaves it out: profiling runs use that flag so that the kernel's trace statistics
    # cover the headline launches only -- this section launches the same kernel on another grid)
    config5 = None if args.no_extras else bench_config5_strong(mfa, rank, world, use_dist)
    traffic = profiled_traffic() if args.workload == "headline" and not args.causal else None

    result = {
        "metric": "attn fwd TFLOP/s & %MFMA peak, B4 H32 N4096 d128 bf16; softmax GB/s",
        "value": round(value, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "priming_launches_untimed": PRIMING,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {torch.bfloat16: "bf16", torch.float16: "f16"}[dtype],
        "data": "synthetic (torch.randn, seed 1 + rank), resident in HBM before timing",
        "config": {"workload": f"attention forward {args.workload}: B={B} H={H} N={N} d={d} per GPU, "
                               f"{'causal' if args.causal else 'non-causal'}, scale={'d^-0.5' if args.scale == 'rsqrt_d' else '1.0'}",
                   "B": B, "H": H, "N": N, "d": d, "sharding": f"batch x heads, {world} rank(s), no collective"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                     # HBM bytes per launch from the committed counter passes of this shape (not collected
                     # by this run: counters need rocprofv3); algorithmic bytes are 4*B*H*N*d*2
                     "traffic": traffic["bytes"] if traffic else None,
                     "traffic_detail": traffic, "algorithmic_bytes": 4.0 * B * H * N * d * 2,
                     "kernel": mfa.attn_fwd_kernel_name(d, dtype, args.causal),
                     "launch_ms_mean": round(mean_ms, 4),
                     "launch_timing": "one HIP event pair around the K timed launches / K", "flops_per_launch": flops},
        "config5_strong_scaling": config5,
    }
    result["verify"] = {"ranks": ranks_seen, "gathered_head_checksums": int(table.shape[0]),
                        "gather_ms_untimed_path": round(gather_ms, 3)}
    if rank == 0 and world == 1 and not args.no_extras and not args.causal:
        result["headline_scale_one"] = bench_shape(mfa, B, H, N, d, dtype, 1.0)
        result["config3"] = bench_shape(mfa, 8, 16, 1024, 64, torch.float16, 64 ** -0.5, iters=100)
        result["config3_scale_one"] = bench_shape(mfa, 8, 16, 1024, 64, torch.float16, 1.0, iters=100)
        result["wide_head_dim"] = bench_shape(mfa, 4, 16, 4096, 256, torch.bfloat16, 256 ** -0.5, iters=10)
        result["wide_head_dim_backward"] = bench_backward(mfa, 4, 16, 4096, 256, torch.bfloat16, 256 ** -0.5)
        result["power_context"] = power_context(mfa, B, H, N, d, dtype, scale)
        result["attention_f32"] = bench_attention_f32(mfa)
        result["attention_backward"] = bench_backward(mfa, B, H, N, d, dtype, scale)
        f32b = bench_backward(mfa, B, H, N, d, torch.float32, scale)
        f32b["frac_f32_mfma_peak_algorithmic"] = round(f32b["TFLOP/s_algorithmic"] / 157.3, 3)
        result["attention_f32_backward"] = f32b
        result["attention_causal"] = bench_causal(mfa, B, H, N, d, dtype, scale)
        result["softmax_sweep"] = bench_softmax_sweep(mfa)
        result["softmax_1024"] = bench_softmax_1024(mfa)
        result["softmax_dim"] = bench_softmax_dim(mfa)
        result["matmul_f32"] = bench_matmul_f32(mfa)
        result["cpu_baseline"] = cpu_baseline(B, H, N, d, scale)
        result["cpu_config1"] = cpu_config1()
    elif rank == 0:
        result["cpu_baseline"] = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
