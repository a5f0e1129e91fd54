#!/bin/bash
set -u
mkdir -p gpurun_out/r3f
o=gpurun_out/r3f
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 14; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi; }
run acc_spec 300 python -u tools/thr_accuracy.py spec=1
run acc_classic 300 python -u tools/thr_accuracy.py spec=0
run acc_spec_f16 300 python -u tools/thr_accuracy.py spec=1 dtype=f16 thrs=0,8,12,15
run ab_scale1 200 python -u tools/ab_attn.py "attn_spec=1,attn_thr=8" "attn_spec=1,attn_thr=12" "attn_spec=1,attn_thr=16" "attn_spec=1,attn_thr=20" "attn_spec=1,attn_thr=24" "attn_spec=1,attn_thr=30" "attn_spec=0,attn_thr=8" "attn_spec=0,attn_thr=4" "attn_spec=0,attn_thr=0" --scale 1.0 --rounds 7
exit 0
