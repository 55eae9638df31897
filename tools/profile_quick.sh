#!/bin/bash
# Kernel trace + the counter passes that say whether a kernel is issue- or latency-bound (fewer passes than
# tools/profile.sh).  Usage: tools/profile_quick.sh <tag> [bench args...]
set -u
TAG=${1:-quick}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-profile --no-legs --no-traffic $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || echo "trace failed"
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  name=$(echo "$ctr" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $ctr failed"
done
find "$OUT" -name "*.csv" | wc -l
