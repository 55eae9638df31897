#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of one workload.
# Usage: tools/profile.sh <tag> [bench args...]          -> gpurun_out/prof_<tag>/...   (profiles bench.py)
#        PROF_CMD="python3 tools/tracer_workload.py" tools/profile.sh <tag>             (profiles another program)
# The program goes straight after `--` (no env / bash -c hop: the profiler's preload has already initialised the GPU).
# Counters are collected in their own passes, never together with a trace (gpurun refuses that combination).
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ -n "${PROF_CMD:-}" ]; then
  BENCH=$(echo "$PROF_CMD" | sed "s#tools/#$ROOT/tools/#g; s#tests/#$ROOT/tests/#g")
else
  BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-profile --no-legs --no-traffic $*"
fi
echo "== kernel trace: $BENCH"; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || echo "trace failed"
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo "$ctr" | tr ' ' '_' | cut -c1-40)
  echo "== pmc $ctr"
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $ctr failed"
done
find "$OUT" -name "*.csv" | wc -l
