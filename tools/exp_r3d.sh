#!/bin/bash
# Round-3 experiment D (GPU box): what the matrix phase costs without its fetches / without a partner.
set -u
mkdir -p gpurun_out/r3d
o=gpurun_out/r3d
E=flash-attention_amd/my_flash_attn_cuda
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 14; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi; }
for f in 0 2 8 16 24 18 1 3; do
run stamp_cvt_s_$f 100 python -u tools/stamp16.py attn_spec=1 attn_xflags=$f lib=$E/exp_cvt.so
done
run stamp_cvt_c_0 100 python -u tools/stamp16.py attn_spec=0 attn_xflags=0 lib=$E/exp_cvt.so
run ab_headline 200 python -u tools/ab_attn.py "attn_spec=1,lib=$E/exp_cvt.so" "attn_spec=1,attn_xflags=1,lib=$E/exp_cvt.so" "attn_spec=1,attn_xflags=2,lib=$E/exp_cvt.so" "attn_spec=1,attn_xflags=3,lib=$E/exp_cvt.so" "attn_spec=0,attn_xflags=2" --rounds 9
exit 0
