#!/bin/bash
# Hardware counters for one kernel of the library: one rocprofv3 --pmc pass per counter group
# (counters only beside the kernel trace), each pass a run of its own, summarised per launch.
#   usage: tools/pmc_cases.sh <tag> <case of tools/pmc_case.py> <kernel-name substring> [more substrings]
set -u
tag=$1; cs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_${tag}_$cs
mkdir -p "$out"
passes=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAVES"
 "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
)
i=0
for p in "${passes[@]}"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$out" -o "p$i" -- python3 tools/pmc_case.py "$cs" > "$out/p$i.log" 2>&1
    rc=$?
    echo "pass $i rc=$rc"
    if [ $rc -ne 0 ]; then tail -5 "$out/p$i.log"; exit $rc; fi
done
grep -h algorithmic "$out/p1.log"
for k in "$@"; do
    echo "--- kernel *$k*"
    python3 tools/pmc_summary.py "$out" "$k"
    python3 tools/trace_kernel_avg.py "$out" "$k"
done
