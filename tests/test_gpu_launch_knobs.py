"""The launch-shaping options (block queue, tail refinement, the short-grid kernel's queue size and stream) change
which workgroup computes a pair and in which order blocks run - never the value of a pair.  Each setting runs on a
fresh set of launches (library.set_option on this thread's context) and must reproduce the default bit for bit."""

import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DEFAULTS = {"persistent": 1, "tail_bpp": 4, "tail_rounds": 1.0, "short_queue": 0, "short_concurrent": 1}


def run_with(options):
    from pyrayhf_amd import library, synth
    for name, value in {**DEFAULTS, **options}.items():
        library.set_option(name, value)
    try:
        alt, den, bmag, bpsi = synth.chapman_profiles(700, 4242)
        freq = synth.sounder_frequencies(4)[::4]
        a = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", 2000)          # 700 blocks > 512 slots
        b = library.vertical_forward_operator(freq, den[:40], bmag[:40], bpsi[:40], alt, "O", 200)
        c = library.vertical_forward_operator(freq, den[:200], bmag[:200], bpsi[:200], alt, "O", 200)   # short-grid kernel
        d = library.vertical_forward_operator_mixed(freq, den, bmag, bpsi, alt,
                                                    [(0, 300, "O", 200), (300, 500, "X", 2000), (500, 700, "O", 500)])
    finally:
        for name, value in DEFAULTS.items():
            library.set_option(name, value)
    return {"a": a, "b": b, "c": c, "d": d}


def test_knobs_do_not_change_results():
    base = run_with({})
    assert np.isfinite(base["a"]).mean() > 0.3 and np.isfinite(base["c"]).mean() > 0.3
    for tag, options in (("no_queue", {"persistent": 0}),
                         ("no_tail", {"tail_bpp": 1}),
                         ("long_tail", {"tail_bpp": 8, "tail_rounds": 0.25}),
                         ("short_queue_overflows", {"short_queue": 16}),
                         ("one_stream", {"short_concurrent": 0})):
        got = run_with(options)
        for key in base:
            assert np.array_equal(got[key], base[key], equal_nan=True), (tag, key)
