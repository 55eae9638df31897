"""The host-side C / C++ of the build under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only -
GPU sanitizers are not available on the pool): the launch planner of libprhf.so (pyrayhf_amd/csrc/prhf_plan.h - the very
code prhf_api.cpp compiles), the double-double sin / cos / pow of the reference-order tier (prhf_crmath.h) and the plain-C
oracle (oracle/vfo_oracle.c), driven by tests/devtools/sanitize_host.cpp.  Any sanitizer report fails the test."""

import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planner_crmath_and_c_oracle_under_asan_and_ubsan(tmp_path):
    if shutil.which("g++") is None or shutil.which("gcc") is None:
        pytest.skip("no gcc / g++")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    inc = ["-I", os.path.join(REPO, "include"), "-I", os.path.join(REPO, "pyrayhf_amd", "csrc"), "-I", "/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__"]          # prhf_kernels.h includes hip_runtime_api.h for its launch prototypes (types only)
    objs = []
    for src, cc, extra in ((os.path.join(REPO, "oracle", "vfo_oracle.c"), "gcc", ["-fopenmp", "-fno-fast-math"]),
                           (os.path.join(REPO, "tests", "devtools", "crmath_host.cpp"), "g++", ["-std=c++17"]),
                           (os.path.join(REPO, "tests", "devtools", "sanitize_host.cpp"), "g++", ["-std=c++17", "-Wall"])):
        obj = tmp_path / (os.path.basename(src) + ".o")
        subprocess.run([cc, *san, "-ffp-contract=off", *extra, *inc, "-c", src, "-o", str(obj)], check=True)
        objs.append(str(obj))
    exe = tmp_path / "sanitize_host"
    subprocess.run(["g++", *san, "-fopenmp", *objs, "-o", str(exe), "-lm"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="2")
    run = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-6000:]
    assert "sanitize_host: ok" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-6000:]
