#!/bin/bash
# Round-3 experiment I (GPU box): resident-workgroup build after the phase-tail changes: parity, then A/B.
set -u
mkdir -p gpurun_out/r3i
o=gpurun_out/r3i
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 8; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ]; then exit $rc; fi; }
run stream_items 600 python -u -m pytest tests/test_variants_gpu.py -m gpu -x -q --timeout 60 --timeout-method thread -k "resident or schedules or mfma_shapes"
run ab_headline 200 python -u tools/ab_attn.py "attn_variant=3,attn_pre=0" "attn_variant=3,attn_pre=1" "attn_variant=2" --rounds 9
run ab_ones 200 python -u tools/ab_attn.py "attn_variant=3,attn_pre=0" "attn_variant=3,attn_pre=1" --rounds 5 --data ones
run ab_cfg5 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3" --shape 1,32,16384,128 --rounds 5 --iters 3
run ab_scale1 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3" --scale 1.0 --rounds 7
run ab_causal 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3" --causal --rounds 7
run ab_d64 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3,attn_pre=0" "attn_variant=3,attn_pre=1" --shape 4,32,4096,64 --rounds 7
run ab_cfg3 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3,attn_pre=0" "attn_variant=3,attn_pre=1" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50
exit 0
