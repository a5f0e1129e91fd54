#!/bin/bash
# Round-3 experiment C (GPU box): phase anatomy (stamps) and conversions before the barrier.
set -u
mkdir -p gpurun_out/r3c
o=gpurun_out/r3c
E=flash-attention_amd/my_flash_attn_cuda
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 14; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi; }
run stamp_c0 100 python -u tools/stamp16.py attn_spec=0
run stamp_c2 100 python -u tools/stamp16.py attn_spec=0 attn_xflags=2
run stamp_s0 100 python -u tools/stamp16.py attn_spec=1
run stamp_s2 100 python -u tools/stamp16.py attn_spec=1 attn_xflags=2
run stamp_c0_ones 100 python -u tools/stamp16.py attn_spec=0 data=ones
run stamp_s0_ones 100 python -u tools/stamp16.py attn_spec=1 data=ones
run ab_headline 200 python -u tools/ab_attn.py "attn_spec=0,attn_xflags=2" "attn_spec=0,attn_xflags=2,lib=$E/exp_cvt.so" "attn_spec=1,attn_xflags=2" "attn_spec=1,attn_xflags=2,lib=$E/exp_cvt.so" "attn_spec=0,lib=$E/exp_cvt.so" "attn_spec=1,lib=$E/exp_cvt.so" "attn_variant=1" --rounds 9
run ab_ones 200 python -u tools/ab_attn.py "attn_spec=0,attn_xflags=2" "attn_spec=0,attn_xflags=2,lib=$E/exp_cvt.so" "attn_spec=1,attn_xflags=2" "attn_spec=1,attn_xflags=2,lib=$E/exp_cvt.so" "attn_variant=1" --rounds 5 --data ones
run ab_d64 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,lib=$E/exp_cvt.so" "attn_spec=1,attn_xflags=2,lib=$E/exp_cvt.so" --shape 4,32,4096,64 --rounds 7
run ab_cfg3 200 python -u tools/ab_attn.py "attn_spec=1" "attn_spec=1,lib=$E/exp_cvt.so" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50
exit 0
