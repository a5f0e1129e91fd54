#!/bin/bash
# Collects hardware counters for the attention kernel, one rocprofv3 --pmc pass per counter group
# (counters only: no trace domains beside the kernel trace), each pass a run of its own.
#   usage: tools/pmc_attn.sh <tag> "<ab_attn spec, e.g. attn_variant=-1>" [extra ab_attn args]
set -u
tag=$1; spec=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p "$out"
passes=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS"
 "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
)
i=0
for p in "${passes[@]}"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$out" -o "p$i" -- python3 tools/ab_attn.py "$spec" --rounds 2 --iters 3 "$@" > "$out/p$i.log" 2>&1
    rc=$?
    echo "pass $i rc=$rc"
    if [ $rc -ne 0 ]; then tail -5 "$out/p$i.log"; exit $rc; fi
done
python3 tools/pmc_summary.py "$out" attn_fwd
