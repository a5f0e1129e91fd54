#!/usr/bin/env python3
"""Pin the oracle against the reference and write tests/golden/.

Run in the development container only (it needs the reference checkout):

    python oracle/make_golden.py [--reference /root/reference]

What it does
1. Loads the reference's ``tests/test_flash_attn.py`` as a module.  That file
   does ``import my_flash_attn_cuda`` at import time (the CUDA extension, which
   cannot exist here); an empty stand-in module of that name is registered
   first so the import of the *Python* functions succeeds.  Nothing from the
   stand-in is ever called.
2. With the reference's seed convention (``torch.manual_seed(1)``, ``randn``,
   tests/test_flash_attn.py:138-139,158-161) on the CPU, evaluates the
   reference's ``_softmax``, ``_attention_forward`` and
   ``_flash_attention_forward`` plus the PyTorch ops its tests compare against
   (``F.softmax``, ``scaled_dot_product_attention(scale=1.0)``) on every shape
   its tests use, plus the CPU reference configuration (1024x1024 softmax, N=128 d=64).
3. Checks the numpy oracle (oracle/ref_cpu.py) and the C oracle
   (oracle/oracle.c) against all of them; aborts if any disagree beyond
   the stated bound.
4. Writes inputs + reference outputs as .npz fixtures and a MANIFEST.json with
   the measured agreements.  The fixtures are data only (arrays); no reference
   source text is stored.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import c_oracle, ref_cpu  # noqa: E402

# shapes the reference's tests use (tests/test_flash_attn.py:133,155,173)
SOFTMAX_SHAPES = [(255, 516), (1024, 1024)]
# the commented-out lists at :134-135 are covered too, as oracle-vs-reference checks only
SOFTMAX_EXTRA = [(256, 256), (99, 99), (100, 2048), (100, 2047)]
ATTN_SHAPES = [(2, 2), (1, 1), (2, 20), (2, 99), (20, 2), (99, 20),
               (1024, 64), (1023, 63), (1025, 65), (128, 64)]

# agreement bounds (float32 reference arithmetic vs float64 oracle)
SOFTMAX_BOUND = 1e-6
ATTN_BOUND = 5e-5   # the reference computes in float32; at N~1024 its own rounding error is ~1e-5


# (B, H, N, d, scale) for the causal-mask fixtures
CAUSAL_SHAPES = [(1, 2, 100, 64, 0.125), (2, 1, 257, 128, 1.0), (1, 1, 64, 40, 0.5)]


# (B, H, N, d, scale, causal) for the backward fixtures
BWD_SHAPES = [(1, 2, 100, 64, 0.125, False), (1, 2, 129, 128, 128 ** -0.5, True), (2, 1, 64, 64, 1.0, True)]


# softmax along a dimension other than the last has no counterpart in the reference: pinned
# against torch.softmax in float64.  (shape, dim)
SOFTMAX_DIM_SHAPES = [((300, 130), 0), ((4, 37, 96), 1), ((2, 3, 50, 7), 2), ((3, 700, 5), 1), ((6, 33, 64), -1)]


def load_reference(ref_root):
    sys.dont_write_bytecode = True
    sys.modules.setdefault("my_flash_attn_cuda", types.ModuleType("my_flash_attn_cuda"))
    path = os.path.join(ref_root, "tests", "test_flash_attn.py")
    spec = importlib.util.spec_from_file_location("_reference_tests", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None, help="regenerate one section only (softmax_dim), keeping the rest of the manifest")
    args = ap.parse_args()
    if args.only in ("softmax_dim", "softmax_fused"):
        with open(os.path.join(args.out, "MANIFEST.json")) as f:
            manifest = json.load(f)
        (softmax_dim_section if args.only == "softmax_dim" else softmax_fused_section)(args, manifest)
        with open(os.path.join(args.out, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    ref = load_reference(args.reference)
    os.makedirs(args.out, exist_ok=True)
    c_oracle.build()
    torch.set_num_threads(8)
    manifest = {"generator": "oracle/make_golden.py", "torch": torch.__version__,
                "seed": 1, "softmax": {}, "attention": {}}

    for (h, w) in SOFTMAX_SHAPES + SOFTMAX_EXTRA:
        torch.manual_seed(1)
        x = torch.randn(h, w, dtype=torch.float32)
        out_ref = ref._softmax(x).numpy()
        out_pt = torch.nn.functional.softmax(x, dim=1).numpy()
        xn = x.numpy()
        o_np = ref_cpu.softmax_rows(xn)
        o_c = c_oracle.softmax_rows(xn)
        o_c32 = c_oracle.softmax_rows(xn, f32_arith=True)
        rec = {
            "ref_vs_torch": maxdiff(out_ref, out_pt),
            "oracle_np_vs_ref": maxdiff(o_np, out_ref),
            "oracle_c_vs_ref": maxdiff(o_c, out_ref),
            "oracle_c_f32_vs_ref": maxdiff(o_c32, out_ref),
            "oracle_np_vs_torch": maxdiff(o_np, out_pt),
            "x_sha256": sha(xn),
        }
        for k in ("oracle_np_vs_ref", "oracle_c_vs_ref", "oracle_c_f32_vs_ref", "oracle_np_vs_torch"):
            assert rec[k] <= SOFTMAX_BOUND, (h, w, k, rec[k])
        name = f"softmax_{h}x{w}"
        if (h, w) in SOFTMAX_SHAPES:
            if h * w > 300_000:      # keep the big one small: every 8th row
                rows = np.arange(0, h, 8)
                np.savez_compressed(os.path.join(args.out, name + ".npz"), x=xn[rows], rows=rows,
                                    out_ref=out_ref[rows], out_torch=out_pt[rows])
                rec["stored_rows"] = int(rows.size)
            else:
                np.savez_compressed(os.path.join(args.out, name + ".npz"), x=xn,
                                    rows=np.arange(h), out_ref=out_ref, out_torch=out_pt)
                rec["stored_rows"] = h
        manifest["softmax"][name] = rec
        print(name, rec)

    for (n, d) in ATTN_SHAPES:
        torch.manual_seed(1)
        q = torch.randn(n, d, dtype=torch.float32)
        k = torch.randn(n, d, dtype=torch.float32)
        v = torch.randn(n, d, dtype=torch.float32)
        o_naive = ref._attention_forward(q, k, v).numpy()
        o_flash = ref._flash_attention_forward(q, k, v).numpy()
        o_sdpa = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1.0).numpy()
        qn, kn, vn = q.numpy(), k.numpy(), v.numpy()
        o_np = ref_cpu.attention_naive(qn, kn, vn)
        o_alg1, _, _ = ref_cpu.attention_flash_alg1(qn, kn, vn)
        o_fa2 = ref_cpu.attention_flash_fa2(qn, kn, vn)
        o_c = c_oracle.attention(qn, kn, vn)
        o_c32 = c_oracle.attention(qn, kn, vn, f32_arith=True)
        rec = {
            "ref_naive_vs_sdpa": maxdiff(o_naive, o_sdpa),
            "ref_flash_vs_sdpa": maxdiff(o_flash, o_sdpa),
            "oracle_np_vs_ref_naive": maxdiff(o_np, o_naive),
            "oracle_alg1_vs_ref_flash": maxdiff(o_alg1, o_flash),
            "oracle_fa2_vs_ref_naive": maxdiff(o_fa2, o_naive),
            "oracle_c_vs_ref_naive": maxdiff(o_c, o_naive),
            "oracle_c_f32_vs_ref_naive": maxdiff(o_c32, o_naive),
            "oracle_np_vs_sdpa": maxdiff(o_np, o_sdpa),
            "q_sha256": sha(qn),
        }
        for key, val in rec.items():
            if key.startswith("oracle_"):
                assert val <= ATTN_BOUND, (n, d, key, val)
        name = f"attn_{n}x{d}"
        np.savez_compressed(os.path.join(args.out, name + ".npz"), q=qn, k=kn, v=vn,
                            out_ref_naive=o_naive, out_ref_flash=o_flash, out_sdpa=o_sdpa)
        manifest["attention"][name] = rec
        print(name, rec)

    # The causal mask has no counterpart in the reference: its oracles are pinned against
    # PyTorch's own scaled_dot_product_attention(is_causal=True), in float64, and the fixtures
    # hold that output.
    manifest["attention_causal"] = {}
    for (b, h, n, d, scale) in CAUSAL_SHAPES:
        torch.manual_seed(1)
        q, k, v = (torch.randn(b, h, n, d, dtype=torch.float32) for _ in range(3))
        o_sdpa = torch.nn.functional.scaled_dot_product_attention(
            q.double(), k.double(), v.double(), is_causal=True, scale=scale).numpy()
        qn, kn, vn = q.numpy(), k.numpy(), v.numpy()
        o_np = ref_cpu.attention_bhnd(qn, kn, vn, scale=scale, causal=True)
        o_c = c_oracle.attention(qn, kn, vn, scale=scale, causal_pos=np.arange(n))
        rec = {"oracle_np_vs_sdpa_f64": maxdiff(o_np, o_sdpa), "oracle_c_vs_sdpa_f64": maxdiff(o_c, o_sdpa),
               "q_sha256": sha(qn)}
        assert rec["oracle_np_vs_sdpa_f64"] <= 1e-12 and rec["oracle_c_vs_sdpa_f64"] <= 1e-6, rec
        name = f"attn_causal_{b}x{h}x{n}x{d}"
        np.savez_compressed(os.path.join(args.out, name + ".npz"), q=qn, k=kn, v=vn,
                            out_sdpa_f64=o_sdpa.astype(np.float32), scale=np.float32(scale))
        manifest["attention_causal"][name] = rec
        print(name, rec)

    # The backward has no counterpart in the reference either: its oracle is pinned against
    # autograd of PyTorch's scaled_dot_product_attention in float64.
    manifest["attention_bwd"] = {}
    for (b, h, n, d, scale, causal) in BWD_SHAPES:
        torch.manual_seed(1)
        q, k, v, do = (torch.randn(b, h, n, d, dtype=torch.float32) for _ in range(4))
        qd, kd, vd = (x.double().requires_grad_(True) for x in (q, k, v))
        o = torch.nn.functional.scaled_dot_product_attention(qd, kd, vd, is_causal=causal, scale=scale)
        o.backward(do.double())
        ref = [x.grad.numpy() for x in (qd, kd, vd)]
        got = ref_cpu.attention_bwd_bhnd(q.numpy(), k.numpy(), v.numpy(), do.numpy(), scale=scale, causal=causal)
        rec = {f"oracle_{nm}_vs_autograd_f64": maxdiff(g, r) for nm, g, r in zip(("dq", "dk", "dv"), got, ref)}
        assert max(rec.values()) <= 1e-10, rec
        rec["q_sha256"] = sha(q.numpy())
        name = f"attn_bwd_{b}x{h}x{n}x{d}" + ("_causal" if causal else "")
        np.savez_compressed(os.path.join(args.out, name + ".npz"), q=q.numpy(), k=k.numpy(), v=v.numpy(), dout=do.numpy(),
                            dq=ref[0].astype(np.float32), dk=ref[1].astype(np.float32), dv=ref[2].astype(np.float32),
                            scale=np.float32(scale), causal=np.bool_(causal))
        manifest["attention_bwd"][name] = rec
        print(name, rec)

    softmax_dim_section(args, manifest)
    softmax_fused_section(args, manifest)

    with open(os.path.join(args.out, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", args.out)


def softmax_dim_section(args, manifest):
    manifest["softmax_dim"] = {}
    for (shape, dim) in SOFTMAX_DIM_SHAPES:
        torch.manual_seed(1)
        x = torch.randn(*shape, dtype=torch.float32) * 3
        o_pt = torch.softmax(x.double(), dim=dim).numpy()
        o_np = ref_cpu.softmax_dim(x.numpy(), dim)
        rec = {"oracle_np_vs_torch_f64": maxdiff(o_np, o_pt), "x_sha256": sha(x.numpy())}
        assert rec["oracle_np_vs_torch_f64"] <= 1e-15, rec
        name = "softmax_dim_" + "x".join(map(str, shape)) + f"_d{dim % len(shape)}"
        np.savez_compressed(os.path.join(args.out, name + ".npz"), x=x.numpy(), dim=np.int64(dim),
                            out_torch_f64=o_pt.astype(np.float32))
        manifest["softmax_dim"][name] = rec
        print(name, rec)


# (shape, dim, scale, mask shape or None, causal): softmax(scale * x + mask) with the mask
# broadcast from the right and -inf planted in it; pinned to torch.softmax in float64
SOFTMAX_FUSED_CASES = [
    ((2, 3, 40, 40), -1, 0.125, (40, 40), False),
    ((2, 2, 33, 33), -1, 0.7, None, True),
    ((3, 50, 50), -1, 1.0, (3, 50, 50), True),
    ((4, 37, 96), 1, 0.5, (37, 96), False),
    ((2, 300, 20), 1, 2.0, (2, 300, 20), False),
    ((5, 64), -1, 0.3, (64,), False),
]


def softmax_fused_section(args, manifest):
    manifest["softmax_fused"] = {}
    for n, (shape, dim, scale, mshape, causal) in enumerate(SOFTMAX_FUSED_CASES):
        torch.manual_seed(10 + n)
        x = torch.randn(*shape, dtype=torch.float32) * 3
        mask = None
        if mshape is not None:
            mask = torch.randn(*mshape, dtype=torch.float32)
            mask[torch.rand(*mshape) < 0.2] = float("-inf")
            # keep one finite entry in every slice along the softmax axis of the mask
            md = dim if dim < 0 else dim - len(shape)
            if -md <= len(mshape):
                idx = [slice(None)] * len(mshape)
                idx[md] = 0
                mask[tuple(idx)] = 0.25
        z = x.double() * scale
        if mask is not None:
            z = z + mask.double()
        if causal:
            nn = shape[-1]
            z = z.masked_fill(torch.ones(nn, nn, dtype=torch.bool).triu(1), float("-inf"))
        o_pt = torch.softmax(z, dim=dim).numpy()
        o_np = ref_cpu.softmax_dim(x.numpy(), dim, scale=scale, mask=None if mask is None else mask.numpy(), causal=causal)
        rec = {"oracle_np_vs_torch_f64": maxdiff(o_np, o_pt), "x_sha256": sha(x.numpy())}
        assert rec["oracle_np_vs_torch_f64"] <= 1e-15, rec
        name = f"softmax_fused_{n}_" + "x".join(map(str, shape))
        np.savez_compressed(os.path.join(args.out, name + ".npz"), x=x.numpy(), dim=np.int64(dim), scale=np.float64(scale),
                            mask=np.zeros(0, np.float32) if mask is None else mask.numpy(), has_mask=np.bool_(mask is not None),
                            causal=np.bool_(causal), out_torch_f64=o_pt.astype(np.float32))
        manifest["softmax_fused"][name] = rec
        print(name, rec)


if __name__ == "__main__":
    main()
