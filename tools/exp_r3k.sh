#!/bin/bash
# Round-3 experiment K (GPU box): causal launches on the pre-scaled speculative path.
set -u
mkdir -p gpurun_out/r3k
o=gpurun_out/r3k
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 8; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ]; then exit $rc; fi; }
run tests 900 python -u -m pytest tests/test_variants_gpu.py tests/test_attention_causal_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "resident or mfma_shapes or causal or fuzz"
run ab_causal 200 python -u tools/ab_attn.py "attn_variant=3,attn_pre=0" "attn_variant=3,attn_pre=1" "attn_variant=2" --causal --rounds 9
run ab_causal16k 200 python -u tools/ab_attn.py "attn_variant=3,attn_pre=0" "attn_variant=3,attn_pre=1" "attn_variant=2" --causal --shape 2,32,16384,128 --rounds 5 --iters 3
run ab_causal_ones 200 python -u tools/ab_attn.py "attn_variant=3,attn_pre=0" "attn_variant=3,attn_pre=1" --causal --rounds 5 --data ones
exit 0
