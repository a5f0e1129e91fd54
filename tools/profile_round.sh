#!/bin/bash
# Produces the evidence kept under profiles/: kernel-trace statistics of the bench's attention
# launches (--no-extras, so that the kernel's average covers the same launches the bench times),
# a plain full bench run, and the PMC passes of the headline attention kernel.  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh <tag>
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p "$out"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --steps 200 --warmup 3 --no-extras > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
rc=$?; echo "bench under rocprofv3 rc=$rc"
[ $rc -ne 0 ] && { tail -5 "$out/bench_under_rocprof.err"; exit $rc; }
python3 tools/trace_timed_avg.py "$out/bench_kernel_trace.csv" 200 > "$out/attn_headline_timed_launches.txt"; cat "$out/attn_headline_timed_launches.txt"
timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
rc=$?; echo "plain bench rc=$rc"
[ $rc -ne 0 ] && { tail -5 "$out/bench.err"; exit $rc; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/full" -o bench -- python3 bench.py > "$out/full_bench_under_rocprof.json" 2> "$out/full_bench_under_rocprof.err"
rc=$?; echo "full bench under rocprofv3 rc=$rc"
[ $rc -ne 0 ] && { tail -5 "$out/full_bench_under_rocprof.err"; exit $rc; }
bash tools/pmc_attn.sh "$tag" "attn_variant=-1" > "$out/attn_headline_pmc.txt" 2>&1
rc=$?; echo "pmc rc=$rc"; tail -3 "$out/attn_headline_pmc.txt"
find "$out" -name "*kernel_stats.csv" | head
exit $rc
