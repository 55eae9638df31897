#!/bin/bash
# Fresh-seed random sweeps on the GPU box (tests/devtools/random_sweep*.py) -> gpurun_out/random_sweep*_<tag>.log
TAG=${1:-r05}; B=${2:-15100}; S=${3:-15000}; N=${4:-400}
mkdir -p gpurun_out
timeout -k 10 420 python tests/devtools/random_sweep_batches.py $B 50 > gpurun_out/random_sweep_batches_$TAG.log 2>&1; echo "batches rc=$?"; tail -2 gpurun_out/random_sweep_batches_$TAG.log
timeout -k 10 300 python tests/devtools/random_sweep.py $S 24 > gpurun_out/random_sweep_$TAG.log 2>&1; echo "small rc=$?"; tail -2 gpurun_out/random_sweep_$TAG.log
timeout -k 10 300 python tests/devtools/random_sweep_nan_tall.py $N 20 > gpurun_out/random_sweep_nan_tall_$TAG.log 2>&1; echo "nan/tall rc=$?"; tail -2 gpurun_out/random_sweep_nan_tall_$TAG.log
