#!/bin/bash
# Round-3 experiment A (GPU box): speculative softmax vs the vote on tile maxima, static priority.
set -u
mkdir -p gpurun_out/r3a
o=gpurun_out/r3a
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; tail -n 12 "$o/$name.log"; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi; }
run variants 600 python -u -m pytest tests/test_variants_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "schedules or mfma_shapes"
run forced 300 python -u -m pytest tests/test_attention_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "rescale or lowp or headline"
run ab_headline 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=1" "attn_spec=1,attn_xflags=1" "attn_spec=0,attn_xflags=1" --rounds 9
run ab_headline_ones 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=1" "attn_spec=1,attn_xflags=1" --rounds 5 --data ones
run ab_scale1 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=0,attn_thr=16" "attn_spec=0,attn_thr=30" "attn_spec=1,attn_thr=8" "attn_spec=1,attn_thr=16" "attn_spec=1" "attn_spec=1,attn_thr=45" --scale 1.0 --rounds 7
run ab_cfg3 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=1" "attn_spec=1,attn_xflags=1" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50
run ab_cfg3_s1 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=1" "attn_spec=1,attn_thr=15" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50 --scale 1.0
run ab_d64 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=1" --shape 4,32,4096,64 --rounds 7
run ab_causal 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=1" "attn_spec=1,attn_xflags=1" --causal --rounds 7
run ab_causal64 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=1" --causal --shape 4,32,4096,64 --rounds 7
exit 0
