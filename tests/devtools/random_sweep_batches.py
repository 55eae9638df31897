"""Randomised sweep over BATCHES big enough (>= 4096 pairs) to take the long-launch paths: candidate list, O-mode
running-maximum search, four-frequency items with shared tails (81 <= n_points <= 1000), main loop with partial
iterations / last grid point / top-segment phase, hint-table variant on non-uniform grids.  X mode against the
plain-C oracle (1e-7), O mode against the NumPy oracle under the per-pair noise rule (floor computed here).
Usage: python tests/devtools/random_sweep_batches.py [first_seed] [n_seeds]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from parity import rel_err, assert_o_mode, oracle_noise
from oracle import vfo_c, vfo_numpy
from pyrayhf_amd import library


def random_batch(rng):
    n_alt = int(rng.integers(40, 300))
    if rng.random() < 0.6:
        alt = 80.0 + np.arange(n_alt) * rng.uniform(0.8, 6.0)
    else:
        alt = 70.0 + np.cumsum(rng.uniform(0.4, 5.0, n_alt))
    P = int(rng.integers(36, 70))
    hm = rng.uniform(alt[n_alt // 3], alt[-1] * 1.05, (P, 1))
    h1 = rng.uniform(15, 80, (P, 1))
    den = 10.0 ** rng.uniform(10.8, 12.6, (P, 1)) * np.exp(0.5 * (1 - (alt - hm) / h1 - np.exp(-(alt - hm) / h1)))
    if rng.random() < 0.6:       # an E layer that leaves a valley
        den = den + 10.0 ** rng.uniform(10.0, 11.4, (P, 1)) * np.exp(-((alt - alt[n_alt // 6]) / rng.uniform(3, 15)) ** 2)
    if rng.random() < 0.3:       # a plateau
        k = int(rng.integers(0, n_alt - 1))
        den[:, k + 1] = den[:, k]
    if rng.random() < 0.2:
        den[:, 0] = 0.0          # vacuum at the bottom
    bmag = rng.uniform(2e-5, 6e-5, (P, 1)) * (1.0 - 3e-4 * (alt - alt[0])) + np.zeros((P, n_alt))
    slope = rng.choice([0.0, 0.0005, 0.002, 0.01, 0.06]) * rng.choice([-1.0, 1.0])
    bpsi = np.clip(rng.uniform(1.0, 89.0, (P, 1)) + slope * (alt - alt[0]), 0.0, 179.0) + np.zeros((P, n_alt))
    if rng.random() < 0.2:
        bpsi = bpsi + np.where(alt > alt[n_alt // 2], rng.uniform(0.1, 2.0), 0.0)
    F = int(rng.integers(115, 260))
    freq = np.sort(rng.uniform(0.4, 15.0, F))
    if rng.random() < 0.3:
        freq = rng.permutation(freq)              # unsorted sweeps are allowed
    n_points = int(rng.choice([81, 100, 137, 200, 256, 333, 500, 1000, 1001, 1024, 1500, 2100]))
    return freq, den, bmag, bpsi, alt, n_points


def check(seed, verbose=True):
    rng = np.random.default_rng(seed)
    freq, den, bmag, bpsi, alt, n = random_batch(rng)
    if np.any(np.argmax(den, axis=1) == 0):
        return 0, 0
    bad = 0
    want_x = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", n)
    for tier in (None, library.MATH_FAITHFUL):
        got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", n, math=tier)
        mask = np.isnan(got) != np.isnan(want_x)
        err, ok = rel_err(got, want_x)
        if mask.any() or err.max(initial=0.0) > 1e-7:
            bad += 1
            print(f"seed {seed} X tier {tier} n_points {n} P {den.shape[0]} F {freq.size}: mask diffs {int(mask.sum())} max err {err.max(initial=0.0):.2e}", flush=True)
    rows = slice(0, 12)                                    # the NumPy oracle and its noise floor on a dozen rows
    with np.errstate(all="ignore"):
        want_o = vfo_numpy.virtual_heights_batch(freq, den[rows], bmag[rows], bpsi[rows], alt, "O", n)
    noise = oracle_noise(freq, den[rows], bmag[rows], bpsi[rows], alt, "O", n, runs=24, seed=seed)   # (8 runs miss a one-ulp flip of pow in ~1 of 200 problems)
    for tier in (None, library.MATH_FAITHFUL):
        got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", n, math=tier)[rows]
        try:
            assert_o_mode(got, want_o, noise, min_within=0.97)
        except AssertionError as exc:
            bad += 1
            print(f"seed {seed} O tier {tier} n_points {n} P {den.shape[0]} F {freq.size}: {exc}", flush=True)
    return 4, bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    checked = bad = 0
    for seed in range(first, first + count):
        c, b = check(seed)
        checked += c; bad += b
        if (seed - first) % 10 == 9:
            print(f"... seed {seed}: {checked} launches checked, {bad} violations", flush=True)
    print(f"done: {checked} launches checked, {bad} violations")
