#!/bin/bash
# Round-3 experiment N (GPU box): head dim 64 on the resident-workgroup build with one workgroup per CU (256 registers).
set -u
mkdir -p gpurun_out/r3n
o=gpurun_out/r3n
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 8; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ]; then exit $rc; fi; }
run ab_cfg3 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3,attn_occ=1,attn_pre=1" "attn_variant=3,attn_occ=1,attn_pre=0" "attn_variant=3,attn_occ=0,attn_pre=0" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50
run ab_d64 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3,attn_occ=1,attn_pre=1" "attn_variant=3,attn_occ=1,attn_pre=0" --shape 4,32,4096,64 --rounds 7
run ab_d64c 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3,attn_occ=1,attn_pre=1" "attn_variant=3,attn_occ=1,attn_pre=0" --shape 4,32,4096,64 --rounds 7 --causal
run ab_cfg3c 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3,attn_occ=1,attn_pre=1" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50 --causal
run ab_cfg3s1 200 python -u tools/ab_attn.py "attn_variant=2" "attn_variant=3,attn_occ=1" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50 --scale 1.0
exit 0
