"""The paths the diagnostic knobs switch back to (one workgroup per block instead of the block queue, no candidate
list, level scan per pair instead of per thread, one workgroup per profile) must stay correct: a subset of the
parity tests in a process of its own with those knobs set (they are read when the library loads)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("knobs", [
    {"PRHF_PERSISTENT": "0", "PRHF_SPLIT_FEW_PROFILES": "0"},
    {"PRHF_NO_CANDIDATES": "1"},
    {"PRHF_THREAD_SCAN_MIN": "1e18", "PRHF_TAIL_BPP": "1"},
])
def test_parity_subset_under_knobs(knobs):
    env = dict(os.environ, **knobs)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
           "-k", "g1 or g5 or between_512 or below_the_gyro", "-p", "no:cacheprovider"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-2000:]
    assert " passed" in done.stdout
