#!/bin/bash
# Everything profiles/ cites for a round, in one gpurun call: bench lines (default with CPU legs, config 3, config 5,
# two-rank rehearsals), cost splits, O-mode reports, rocprof trace + PMC passes of every kernel.
# Usage (on the GPU box, from the repo root): tools/final_measurements.sh r02
TAG=${1:-r02}
OUT=gpurun_out/final_$TAG; mkdir -p $OUT
python bench.py > $OUT/bench_${TAG}_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
python bench.py --workload config3 --steps 20 --no-single-profile > $OUT/bench_${TAG}_config3.json 2> $OUT/bench_c3.err; echo "config3 rc=$?"
python bench.py --workload config5 --steps 10 --no-single-profile > $OUT/bench_${TAG}_config5.json 2> $OUT/bench_c5.err; echo "config5 rc=$?"
PRHF_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --no-cpu-baseline > $OUT/bench_${TAG}_n2_rehearsal.json 2> $OUT/bench_n2.err; echo "n2 rc=$?"
PRHF_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload config5 --steps 3 --no-cpu-baseline > $OUT/bench_${TAG}_config5_n2_rehearsal.json 2> $OUT/bench_c5n2.err; echo "c5 n2 rc=$?"
python tools/stage_cost3.py > $OUT/stage_cost3_$TAG.jsonl 2>/dev/null
python tools/stage_cost.py > $OUT/stage_cost4_$TAG.jsonl 2>/dev/null
python tools/bench_generic_path.py > $OUT/bench_generic_path_$TAG.jsonl 2>/dev/null
python tools/tracer_fan_workload.py > $OUT/bench_tracer_fan_$TAG.jsonl 2>/dev/null
python tests/devtools/omode_report.py > $OUT/omode_report_$TAG.jsonl 2>/dev/null
python tests/devtools/omode_sweep.py > $OUT/omode_sweep_$TAG.jsonl 2>/dev/null
tools/profile.sh ${TAG}_config4 > /dev/null 2>&1
tools/profile.sh ${TAG}_config3 --workload config3 > /dev/null 2>&1
tools/profile.sh ${TAG}_config5 --workload config5 > /dev/null 2>&1
tools/profile.sh ${TAG}_config2 --profiles 1 --freqs 174 > /dev/null 2>&1
PROF_CMD="python3 tools/tracer_workload.py" tools/profile.sh ${TAG}_snell > /dev/null 2>&1
echo "final measurements done"
