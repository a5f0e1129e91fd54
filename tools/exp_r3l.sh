#!/bin/bash
# Round-3 experiment L (GPU box): mask as VCC compare/select pairs (no scalar spills): parity, then A/B against an older experiment build.
set -u
mkdir -p gpurun_out/r3l
o=gpurun_out/r3l
E=flash-attention_amd/my_flash_attn_cuda/exp_${1:-addr}.so
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 8; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ]; then exit $rc; fi; }
run tests 900 python -u -m pytest tests/test_variants_gpu.py tests/test_attention_causal_gpu.py tests/test_fuzz_gpu.py tests/test_attention_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread
run ab_headline 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,lib=$E" "attn_variant=2" --rounds 9
run ab_scale1 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,lib=$E" --scale 1.0 --rounds 7
run ab_causal 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,attn_pre=0" "attn_variant=3,lib=$E" "attn_variant=2" --causal --rounds 9
run ab_causal16k 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,attn_pre=0" "attn_variant=2" --causal --shape 2,32,16384,128 --rounds 5 --iters 3
run ab_ragged 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,lib=$E" "attn_variant=2" --shape 4,32,4000,128 --rounds 7
exit 0
