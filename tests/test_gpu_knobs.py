"""The paths the context options switch back to (one workgroup per block instead of the block queue, no candidate
list, level scan per pair instead of per thread, one workgroup per profile, short O-mode grids in the general kernel,
a short-grid queue that overflows) must stay correct: a subset of the parity tests in a process of its own with those
options set (tests/conftest.py applies PRHF_TEST_OPTIONS through library.set_option)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("knobs", [
    "persistent=0,split_few_profiles=0",
    "no_candidates=1",
    "thread_scan_min=1e18,tail_bpp=1",
    "short_kernel=0",
    "short_queue=24",
    "local_chunks=0,direct_upload=0,timing=1",          # round 3: chunk sums through scratch + a second kernel, staged upload
    "trim_lds=0,shortx_kernel=0,tall_lean=0",           # profiles staged in global memory through the generic loop
    "short_lanes=16,short_order=0,short_prio=1",        # round 5: sixteen lanes per pair at every grid size, index order, priorities by age
])
def test_parity_subset_under_knobs(knobs):
    env = dict(os.environ, PRHF_TEST_OPTIONS=knobs)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
           "-k", "g1 or g4 or g5 or g10 or g13 or between_512 or below_the_gyro", "-p", "no:cacheprovider"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-2000:]
    assert " passed" in done.stdout
