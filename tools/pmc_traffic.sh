#!/bin/bash
# HBM traffic of the bench workload only (two PMC passes): tools/pmc_traffic.sh <tag> [bench args]; env passes through.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/$ctr -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-profile --no-legs --no-traffic $* > $OUT/$ctr.log 2>&1
  python3 - <<PY
import csv, glob
vals=[float(r["Counter_Value"]) for p in glob.glob("$OUT/$ctr/*/*_counter_collection.csv") for r in csv.DictReader(open(p)) if "vfo_kernel" in r["Kernel_Name"]]
print("$TAG", "$ctr", sum(vals)/max(len(vals),1), "KiB per dispatch over", len(vals))
PY
done
