#!/usr/bin/env python3
"""Bandwidth of softmax along a non-last dimension (8 bytes per float32 element algorithmic),
next to torch.softmax on the same tensors."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
import my_flash_attn_cuda as mfa

CASES = [((8192, 8192), 0, torch.float32), ((16384, 16384), 0, torch.float32), ((64, 128, 16384), 1, torch.float32),
         ((32, 1024, 4096), 1, torch.float32), ((256, 64, 64, 64), 1, torch.float32), ((4096, 64, 512), 1, torch.float32),
         ((8192, 8192), 0, torch.bfloat16), ((64, 128, 16384), 1, torch.bfloat16)]

# extra cases on the command line: 8192x1024:0:bfloat16
_DT = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}
if len(sys.argv) > 1:
    CASES = [(tuple(int(n) for n in a.split(":")[0].split("x")), int(a.split(":")[1]), _DT[a.split(":")[2]]) for a in sys.argv[1:]]


def med(fn, it=20):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[it // 2]


for shape, dim, dt in CASES:
    xs = [torch.randn(*shape, device="cuda").to(dt) for _ in range(3)]     # rotate: defeat the Infinity Cache
    i = [0]
    def ours():
        i[0] += 1
        return mfa.softmax(xs[i[0] % 3], dim=dim)
    def theirs():
        i[0] += 1
        return torch.softmax(xs[i[0] % 3], dim=dim)
    nbytes = 2 * xs[0].numel() * xs[0].element_size()
    t_o, t_t = med(ours), med(theirs)
    print(f"{str(shape):24s} dim {dim} {str(dt)[6:]:9s} ours {t_o:8.3f} ms {nbytes / t_o / 1e9:7.2f} TB/s   torch {t_t:8.3f} ms {nbytes / t_t / 1e9:7.2f} TB/s", flush=True)
