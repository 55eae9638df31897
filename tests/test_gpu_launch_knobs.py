"""The launch-shaping knobs (block queue, tail refinement, chunk target) change which workgroup computes a
pair and in which order blocks run - never the value of a pair.  Each setting runs in a child process
(the knobs are read when the library creates its first context) and must reproduce the default bit for bit."""

import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from pyrayhf_amd import library, synth
alt, den, bmag, bpsi = synth.chapman_profiles(700, 4242)
freq = synth.sounder_frequencies(4)[::4]
a = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", 2000)          # 700 blocks > 512 slots
b = library.vertical_forward_operator(freq, den[:40], bmag[:40], bpsi[:40], alt, "O", 200)
np.savez(sys.argv[1], a=a, b=b)
"""


def run_child(tmp_path, tag, env):
    out = str(tmp_path / f"{tag}.npz")
    full = dict(os.environ, **env)
    subprocess.run([sys.executable, "-c", CHILD % ROOT, out], env=full, check=True, timeout=300)
    return np.load(out)


def test_knobs_do_not_change_results(tmp_path):
    base = run_child(tmp_path, "default", {})
    assert np.isfinite(base["a"]).mean() > 0.3
    for tag, env in (("no_queue", {"PRHF_PERSISTENT": "0"}),
                     ("no_tail", {"PRHF_TAIL_BPP": "1"}),
                     ("long_tail", {"PRHF_TAIL_BPP": "8", "PRHF_TAIL_ROUNDS": "0.25"})):
        got = run_child(tmp_path, tag, env)
        for key in ("a", "b"):
            assert np.array_equal(got[key], base[key], equal_nan=True), (tag, key)
