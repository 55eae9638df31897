set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/h_tests.log 2>&1 || { tail -30 gpurun_out/h_tests.log; exit 1; }
tail -3 gpurun_out/h_tests.log
python bench.py --workload config3 --steps 20 --warmup 3 --no-cpu-baseline --no-single-profile > gpurun_out/c_bench3.json 2> gpurun_out/c_bench.err
python bench.py --workload config5 --steps 10 --warmup 2 --no-cpu-baseline --no-single-profile > gpurun_out/c_bench5.json 2>> gpurun_out/c_bench.err
python - <<'PY'
import json
for f in ("c_bench3","c_bench5"):
    d=json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["n_gpus"], d["value"], d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"])
PY
