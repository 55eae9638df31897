#!/bin/bash
# Run on the GPU box: tools/f_row_workload.py under a kernel trace and under counters of their own passes.
# Usage: tools/f_row_profile.sh <tag> -> gpurun_out/frow_<tag>/ ; then python tools/f_row_bounds.py gpurun_out/frow_<tag> <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/frow_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PROG="python3 $ROOT/tools/f_row_workload.py"
$PROG > "$OUT/plain.jsonl" 2> "$OUT/plain.log" || echo "plain run failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $PROG > "$OUT/trace.log" 2>&1 || echo "trace failed"
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo "$ctr" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$name" -- $PROG > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $ctr failed"
done
find "$OUT" -name "*.csv" | wc -l
