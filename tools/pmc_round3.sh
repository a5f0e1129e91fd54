#!/bin/bash
# Counter passes for the kernels beside the headline forward (GPU box).  Each case ends the script on failure.
set -u
t=${1:-r03}
only=${2:-}
o=gpurun_out/pmc_$t${only:+_$only}.txt
: > $o
run() { if [ -n "$only" ] && [[ ",$only," != *",$1,"* ]]; then return 0; fi; echo "=== $1 ===" | tee -a $o; bash tools/pmc_cases.sh $t "$@" >> $o 2>&1; rc=$?; tail -3 $o; if [ $rc -ne 0 ]; then echo "FAILED rc=$rc"; exit $rc; fi; }
run softmax32768 softmax_rows
run softmax8192 softmax_rows
run cols8192 softmax_cols
run config3 attn_fwd_
run wide256 attn_fwd_mfma_wide
run causal attn_fwd_
run scale1 attn_fwd_
run bwd attn_bwd_dq attn_bwd_dkv attn_bwd_delta
run bwdwide attn_bwd_dq_wide attn_bwd_dkv_wide
run bwdf32 attn_bwd_f32_kernel
run bwdf32causal attn_bwd_f32_kernel
exit 0
