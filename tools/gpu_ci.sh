#!/bin/bash
# Runs the GPU checks step by step on the GPU box, each under its own time limit.
# A step that merely fails its assertions (exit 1) lets the next step run; anything
# else (time limit, abort, segfault, GPU fault) ends the script there.
#   usage: tools/gpu_ci.sh [step ...]     steps: smoke softmax attention bench
set -u
mkdir -p gpurun_out
steps=("$@")
[ ${#steps[@]} -eq 0 ] && steps=(smoke softmax attention bench)
rc_all=0
run() {  # name, seconds, command...
    local name=$1 secs=$2; shift 2
    echo "=== $name ==="
    timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    tail -n 25 "gpurun_out/$name.log"
    echo "=== $name exit $rc ==="
    if [ $rc -ne 0 ]; then rc_all=$rc; fi
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: $name ended abnormally"; exit $rc; fi
}
for s in "${steps[@]}"; do
    case $s in
        smoke)     run smoke 300 python -u __graft_entry__.py smoke ;;
        softmax)   run softmax 600 python -u -m pytest tests/test_softmax_gpu.py -m gpu -v --timeout 120 --timeout-method thread -x -q ;;
        attention) run attention 900 python -u -m pytest tests/test_attention_gpu.py -m gpu -v --timeout 180 --timeout-method thread -q ;;
        all)       run all 1100 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread ;;
        bench)     run bench 600 python -u bench.py ;;
        *) echo "unknown step $s"; exit 2 ;;
    esac
done
exit $rc_all
