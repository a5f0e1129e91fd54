#!/bin/bash
# GPU box: the backward tests (all head dims, the allocation-free entry), then the wide-head backward's timing
mkdir -p gpurun_out/bwdw
timeout -k 10 900 python -u -m pytest tests/test_attention_bwd_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread > gpurun_out/bwdw/all.log 2>&1
rc=$?; tail -4 gpurun_out/bwdw/all.log; [ $rc -ne 0 ] && exit $rc
for sh in 4,16,4096,256 4,16,4096,192; do timeout -k 10 200 python tools/bwd_bench.py --shape $sh --no-torch || exit 1; done
timeout -k 10 200 python tools/bwd_bench.py --shape 4,16,4096,256 --causal --no-torch
