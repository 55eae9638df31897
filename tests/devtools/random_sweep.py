"""Wider randomised sweep than tests/test_gpu_random.py (more seeds, default arithmetic included): prints every
violation of the same acceptance rules.  Usage: python tests/devtools/random_sweep.py [first_seed] [n_seeds]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_gpu_random import random_problem
from parity import rel_err
from oracle import vfo_c
from pyrayhf_amd import library
first = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = checked = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    for it in range(40):
        freq, den, bmag, bpsi, alt, n_points = random_problem(rng)
        if np.any(np.argmax(den, axis=1) == 0):
            continue
        for mode in "OX":
            want = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, mode, n_points)
            for tier in (None, library.MATH_FAITHFUL, library.MATH_FAST):
                got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points, math=tier)
                checked += 1
                mask = np.isnan(got) != np.isnan(want)
                err, ok = rel_err(got, want)
                lim = 1e-7 if mode == "X" else 5e-3
                med_bad = mode == "O" and ok.sum() >= 10 and np.median(err[ok]) > 1e-6
                if mask.any() or err.max(initial=0.0) > lim or med_bad:
                    bad += 1
                    print(f"seed {seed} it {it} mode {mode} tier {tier} n_points {n_points}: mask diffs {int(mask.sum())} "
                          f"max err {err.max(initial=0.0):.2e}", flush=True)
    if (seed - first) % 10 == 9:
        print(f"... seed {seed}: {checked} launches checked, {bad} violations", flush=True)
print(f"done: {checked} launches checked, {bad} violations")
