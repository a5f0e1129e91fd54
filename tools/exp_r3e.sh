#!/bin/bash
# Round-3 experiment E (GPU box): row sums of the rounded weights (v_dot2) -- speed and parity.
set -u
mkdir -p gpurun_out/r3e
o=gpurun_out/r3e
E=flash-attention_amd/my_flash_attn_cuda
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 14; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi; }
run ab_headline 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,lib=$E/exp_cvt.so" "attn_spec=0" "attn_spec=0,lib=$E/exp_cvt.so" --rounds 9
run ab_ones 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,lib=$E/exp_cvt.so" --rounds 5 --data ones
run ab_scale1 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,attn_thr=16" "attn_spec=1,attn_thr=45" "attn_spec=1,lib=$E/exp_cvt.so" "attn_spec=0,lib=$E/exp_cvt.so" "attn_spec=0,attn_xflags=2,lib=$E/exp_cvt.so" --scale 1.0 --rounds 7
run ab_cfg3 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,lib=$E/exp_cvt.so" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50
run ab_d64 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,lib=$E/exp_cvt.so" --shape 4,32,4096,64 --rounds 7
run ab_causal 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,lib=$E/exp_cvt.so" "attn_spec=0,lib=$E/exp_cvt.so" --causal --rounds 7
run variants 600 python -u -m pytest tests/test_variants_gpu.py -m gpu -q --timeout 120 --timeout-method thread -k "schedules or mfma_shapes"
run attention 600 python -u -m pytest tests/test_attention_gpu.py tests/test_attention_causal_gpu.py tests/test_fuzz_gpu.py -m gpu -q --timeout 180 --timeout-method thread
exit 0
