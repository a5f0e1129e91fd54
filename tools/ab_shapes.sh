#!/bin/bash
# A/B of the two MFMA shapes of the attention forward over the benchmark shapes (GPU box).
set -u
mkdir -p gpurun_out
V="attn_variant=1 attn_variant=2"
run() { echo "== $*"; timeout -k 10 200 python -u tools/ab_attn.py $V "$@" 2>&1 | grep -v amdgpu.ids || exit 1; }
run --shape 4,32,4096,128
run --shape 4,32,4096,128 --scale 1.0
run --shape 4,32,4096,128 --causal
run --shape 1,32,16384,128 --rounds 3 --iters 3
run --shape 4,32,4096,64
run --shape 8,16,1024,64 --dtype f16 --iters 50
run --shape 8,16,1024,64 --dtype f16 --iters 50 --causal
run --shape 4,32,4096,128 --dtype f16
