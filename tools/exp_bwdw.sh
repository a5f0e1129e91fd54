#!/bin/bash
# GPU box: the wide-head backward tests, then the whole backward file
mkdir -p gpurun_out/bwdw
timeout -k 10 600 python -u -m pytest tests/test_attention_bwd_gpu.py -x -q --timeout 180 --timeout-method thread -k "wide" > gpurun_out/bwdw/wide.log 2>&1
rc=$?; tail -15 gpurun_out/bwdw/wide.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -u -m pytest tests/test_attention_bwd_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread > gpurun_out/bwdw/all.log 2>&1
rc=$?; tail -5 gpurun_out/bwdw/all.log; exit $rc
