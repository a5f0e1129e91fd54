#!/bin/bash
# Per-kernel times of the attention backward at the headline shape (rocprofv3 kernel trace).
# usage: bash tools/bwd_prof.sh <tag> [extra args of tools/bwd_bench.py]
set -e
tag=${1:-bwd}; shift || true
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bwd -- python3 $GRAFT_REPO_ROOT/tools/bwd_bench.py "$@" > $out/bench.log 2>&1
grep -v "^[WE]2026" $out/bench.log | tail -4
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "attn_" in n:
            print(f'{n[:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f} us')
PY
