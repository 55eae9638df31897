#!/bin/bash
# Run on the GPU box: tools/inst_budget3.py under a kernel trace and under instruction counters (own passes).
# Usage: tools/inst_budget3.sh <tag>   -> gpurun_out/budget3_<tag>/ ; then tools/summarize_budget3.py <dir> <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/budget3_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PROG="python3 $ROOT/tools/inst_budget3.py"
$PROG > "$OUT/variants.jsonl" 2> "$OUT/plain.log" || echo "plain run failed"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- $PROG > "$OUT/trace.log" 2>&1 || echo "trace failed"
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo "$ctr" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$name" -- $PROG > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $ctr failed"
done
find "$OUT" -name "*.csv" | wc -l
