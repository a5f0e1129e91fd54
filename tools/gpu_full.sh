#!/bin/bash
# GPU box: every GPU test, then the default bench run (logs under gpurun_out/full).
set -u
mkdir -p gpurun_out/full
o=gpurun_out/full
run() { local name=$1 secs=$2; shift 2; echo "=== $name ==="; timeout -k 10 "$secs" "$@" > "$o/$name.log" 2>&1; local rc=$?; grep -v amdgpu.ids "$o/$name.log" | tail -n 8; echo "=== $name exit $rc ==="; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi; }
run all_gpu 1100 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread -x
run bench 600 python -u bench.py
exit 0
