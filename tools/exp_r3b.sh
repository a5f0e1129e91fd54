#!/bin/bash
# Round-3 experiment B (GPU box): priority of the wave in its matrix phase, speculative softmax or not.
set -u
mkdir -p gpurun_out/r3b
o=gpurun_out/r3b
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; tail -n 12 "$o/$name.log"; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi; }
run ab_headline 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=0,attn_xflags=2" "attn_spec=0,attn_xflags=3" "attn_spec=1" "attn_spec=1,attn_xflags=2" "attn_spec=1,attn_xflags=3" "attn_spec=1,attn_xflags=4" "attn_variant=1" --rounds 9
run ab_headline_ones 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=0,attn_xflags=2" "attn_spec=1" "attn_spec=1,attn_xflags=2" "attn_spec=1,attn_xflags=4" --rounds 5 --data ones
run ab_scale1 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=0,attn_xflags=2" "attn_spec=0,attn_thr=12,attn_xflags=2" "attn_spec=1,attn_xflags=2" "attn_spec=1,attn_thr=16,attn_xflags=2"  --scale 1.0 --rounds 7
run ab_cfg3 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=0,attn_xflags=2" "attn_spec=1" "attn_spec=1,attn_xflags=2" "attn_spec=1,attn_xflags=3" --shape 8,16,1024,64 --dtype f16 --rounds 9 --iters 50
run ab_d64 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=0,attn_xflags=2" "attn_spec=1" "attn_spec=1,attn_xflags=2" --shape 4,32,4096,64 --rounds 7
run ab_causal 200 python -u tools/ab_attn.py "attn_spec=0" "attn_spec=0,attn_xflags=2" "attn_spec=1" "attn_spec=1,attn_xflags=2" --causal --rounds 7
exit 0
