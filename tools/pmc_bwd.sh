#!/bin/bash
# Hardware counters of the backward kernels at the headline shape, one rocprofv3 --pmc pass per
# counter group (counters only, beside the kernel trace), each pass a run of its own.
#   usage: tools/pmc_bwd.sh <tag> <kernel-name substring> [extra args of tools/bwd_bench.py]
set -u
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p "$out"
passes=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS"
 "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES"
)
i=0
for p in "${passes[@]}"; do
    i=$((i+1))
    timeout -s KILL 150 rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$out" -o "p$i" -- python3 tools/bwd_bench.py --no-torch "$@" > "$out/p$i.log" 2>&1
    rc=$?
    echo "pass $i rc=$rc"
    if [ $rc -ne 0 ]; then tail -5 "$out/p$i.log"; exit $rc; fi
done
python3 tools/pmc_summary.py "$out" "$kern"
