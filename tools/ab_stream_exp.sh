#!/bin/bash
# GPU box: in-tree tuning library against an experiment build (my_flash_attn_cuda/exp_<name>.so) of variant 3.
set -u
name=${1:-addr}
mkdir -p gpurun_out/ab_stream
o=gpurun_out/ab_stream
E=flash-attention_amd/my_flash_attn_cuda/exp_$name.so
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 8; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ]; then exit $rc; fi; }
run ab_headline 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,lib=$E" "attn_variant=2" --rounds 11
run ab_ones 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,lib=$E" --rounds 5 --data ones
run ab_scale1 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,lib=$E" --scale 1.0 --rounds 7
run ab_scale1_spec 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,attn_spec=1,attn_pre=0,lib=$E" "attn_variant=3,attn_spec=1,attn_pre=0,attn_thr=60,lib=$E" --scale 1.0 --rounds 7
run ab_causal 200 python -u tools/ab_attn.py "attn_variant=3" "attn_variant=3,lib=$E" --causal --rounds 7
run ab_d64 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3" "attn_variant=3,lib=$E" --shape 4,32,4096,64 --rounds 7
exit 0
