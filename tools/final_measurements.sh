#!/bin/bash
# Everything profiles/ cites for a round, in one or two gpurun calls: bench lines (default with legs and CPU legs,
# config 3, config 5, two-rank rehearsal, one-rank RCCL), cost splits, O-mode reports, rocprof trace + PMC passes.
# Usage (on the GPU box, from the repo root): tools/final_measurements.sh r03 [part]     part: bench | profiles | all
TAG=${1:-r05}
PART=${2:-all}
OUT=gpurun_out/final_$TAG; mkdir -p $OUT
if [ "$PART" = "bench" ] || [ "$PART" = "all" ]; then
python bench.py > $OUT/bench_${TAG}_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
python bench.py --workload config3 --steps 20 --no-single-profile > $OUT/bench_${TAG}_config3.json 2> $OUT/bench_c3.err; echo "config3 rc=$?"
python bench.py --workload config5 --steps 10 --no-single-profile > $OUT/bench_${TAG}_config5.json 2> $OUT/bench_c5.err; echo "config5 rc=$?"
python bench.py --gpus 1 --force-collective --steps 5 --no-cpu-baseline --no-single-profile --no-legs > $OUT/bench_${TAG}_n1_rccl.json 2> $OUT/bench_rccl.err; echo "rccl rc=$?"
PRHF_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --no-cpu-baseline > $OUT/bench_${TAG}_n2_rehearsal.json 2> $OUT/bench_n2.err; echo "n2 rc=$?"
# more ranks on the one GPU (the box admits six processes on the card and the launcher holds it open too - a six-rank
# run was killed by the process guard: four ranks here, N = 8 itself is rehearsed on CPU, tests/test_dist_gloo.py)
PRHF_BENCH_BACKEND=gloo python bench.py --gpus 4 --profiles 256 --steps 3 --no-cpu-baseline --no-single-profile --no-legs > $OUT/bench_${TAG}_n4_rehearsal.json 2> $OUT/bench_n4.err; echo "n4 rc=$?"
PRHF_BENCH_BACKEND=gloo python bench.py --gpus 4 --workload config5 --profiles 256 --steps 3 --no-cpu-baseline --no-single-profile --no-legs > $OUT/bench_${TAG}_config5_n4_rehearsal.json 2> $OUT/bench_c5n4.err; echo "c5 n4 rc=$?"
python tools/tracer_workload.py > $OUT/bench_tracers_$TAG.jsonl 2>/dev/null
python tools/tracer_fan_workload.py > $OUT/bench_tracer_fan_$TAG.jsonl 2>/dev/null
python bench.py --devices-all-child 1 > $OUT/devices_all_one_gpu_$TAG.json 2>/dev/null; echo "devices=all child rc=$?"
python tools/stage_cost3.py > $OUT/stage_cost3_$TAG.jsonl 2>/dev/null
python tools/bench_generic_path.py > $OUT/bench_generic_path_$TAG.jsonl 2>/dev/null
python tools/time_dropin.py > $OUT/time_dropin_$TAG.jsonl 2>/dev/null
python tools/slice_cost5.py > $OUT/slice_cost5_$TAG.jsonl 2>/dev/null
python tests/devtools/omode_report.py > $OUT/omode_report_$TAG.jsonl 2>/dev/null
python tests/devtools/omode_sweep.py > $OUT/omode_sweep_$TAG.jsonl 2>/dev/null
echo "bench part done"
fi
if [ "$PART" = "profiles" ] || [ "$PART" = "all" ]; then
tools/profile.sh ${TAG}_config4 > /dev/null 2>&1; echo "prof config4 done"
tools/profile.sh ${TAG}_config3 --workload config3 > /dev/null 2>&1; echo "prof config3 done"
tools/profile.sh ${TAG}_config5 --workload config5 > /dev/null 2>&1; echo "prof config5 done"
tools/profile.sh ${TAG}_config2 --profiles 1 --freqs 174 > /dev/null 2>&1; echo "prof config2 done"
tools/f_row_profile.sh ${TAG} > /dev/null 2>&1; echo "prof tracers (f-row legs) done"
tools/inst_budget3.sh ${TAG} > /dev/null 2>&1; echo "config 3 instruction budget done"
fi
echo "final measurements done"
