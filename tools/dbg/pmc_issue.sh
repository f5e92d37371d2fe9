#!/bin/bash
# Instruction-issue counters of the closest-hit kernel (two passes of 8 SQ counters): which pipes its cycles go to.
#   tools/dbg/pmc_issue.sh <tag>     -> gpurun_out/issue_<tag>_{a,b}.txt
set -e -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
PMCB="python3 $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
pass() {
    local name=$1
    shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/issue_$name" -o pmc -- $PMCB > "$OUT/issue_$name.log" 2>&1
    find "$OUT/issue_$name" -name '*counter_collection.csv' -exec cp {} "$OUT/issue_$name.csv" \;
    rm -rf "$OUT/issue_$name"
    python3 tools/pmc_summary.py "$OUT/issue_$name.csv" k_trace > "$OUT/issue_$name.txt"
    rm -f "$OUT/issue_$name.csv"
    tail -9 "$OUT/issue_$name.txt"
}
pass ${TAG}_a SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
pass ${TAG}_b SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE
