#!/usr/bin/env python3
"""Re-run one seed of test_random_problems_with_nan_inputs / test_random_tall_columns verbosely (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_random as t
from oracle import vfo_numpy, vfo_c
from pyrayhf_amd import library
from parity import rel_err, oracle_noise
np.seterr(all="ignore")
np.set_printoptions(linewidth=200, precision=6)
kind, seed = sys.argv[1], int(sys.argv[2])
if kind == "nan":
    rng = np.random.default_rng(7000 + seed)
    for it in range(60):
        freq, den, bmag, bpsi, alt, n_points = t.random_problem(rng)
        if np.any(np.argmax(den, axis=1) == 0):
            continue
        alt = np.array(alt, dtype=np.float64, copy=True)
        n_prof, n_alt = den.shape
        victim = int(rng.integers(n_prof))
        what = rng.choice(["den", "alt", "bmag", "bpsi", "bpsi", "bmag"])
        where = None
        if what == "den":
            first = int(rng.integers(1, n_alt)); den[victim, first:] = np.nan; where = first
        else:
            col = {"alt": alt if alt.ndim == 2 else None, "bmag": bmag, "bpsi": bpsi}[what]
            if col is None:
                where = int(rng.integers(n_alt)); alt[where] = np.nan
            else:
                where = rng.integers(0, n_alt, int(rng.integers(1, 4))); col[victim, where] = np.nan
        for mode in "XO":
            want = vfo_numpy.virtual_heights_batch(freq, den, bmag, bpsi, alt, mode, n_points)
            got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points)
            err_, ok_ = rel_err(got, want)
            big = ok_ & np.isfinite(got) & (err_ > (1e-7 if mode == "X" else 3e-6))
            if big.any() and np.array_equal(np.isnan(got), np.isnan(want)):
                noise = oracle_noise(freq, den, bmag, bpsi, alt, mode, n_points, runs=12, seed=seed) if mode == "O" else np.zeros_like(want)
                gotf = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points, math=library.MATH_FAITHFUL)
                for idx in np.argwhere(big):
                    i, j = idx
                    if err_[i, j] > max(1e-6, 4 * noise[i, j]) or mode == "X":
                        print("iteration", it, "mode", mode, "what", what, "victim", victim, "where", where, "n_alt", n_alt, "n_points", n_points,
                              "pair", idx, "f", freq[j], "err", err_[i, j], "noise", noise[i, j], "want", want[i, j], "got", got[i, j], "faithful tier", gotf[i, j])
                        print("peaks", np.argmax(np.where(np.isnan(den), np.inf, den), axis=1))
                        print("den", den[i]); print("bmag", bmag[i]); print("bpsi", bpsi[i]); print("alt", alt if alt.ndim == 1 else alt[i])
            if not np.array_equal(np.isnan(got), np.isnan(want)):
                print("iteration", it, "mode", mode, "what", what, "victim", victim, "where", where, "n_alt", n_alt, "n_points", n_points,
                      "n_prof", n_prof, "alt2d", alt.ndim == 2)
                p = victim
                print("peaks", np.argmax(np.where(np.isnan(den), np.inf, den), axis=1))
                print("freq", freq)
                for q in range(n_prof):
                    print("profile", q, "want", want[q]); print("profile", q, "got ", got[q])
                print("den[p]", den[p]); print("bmag[p]", bmag[p]); print("bpsi[p]", bpsi[p]); print("alt", alt if alt.ndim == 1 else alt[p])
                sys.exit(0)
else:
    rng = np.random.default_rng(8000 + seed)
    for it in range(14):
        freq, den, bmag, bpsi, alt, n_points = t.random_problem(rng)
        if alt.ndim == 2:
            alt = alt[0]
        n_tall = int(rng.integers(1401, 4000))
        fine = np.linspace(alt[0], alt[-1], n_tall)
        den, bmag, bpsi = (np.array([np.interp(fine, alt, r) for r in x]) for x in (den, bmag, bpsi))
        if np.any(np.argmax(den, axis=1) == 0):
            continue
        n_points = min(n_points, 777)
        want_o = vfo_numpy.virtual_heights_batch(freq, den, bmag, bpsi, fine, "O", n_points)
        want_c = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, fine, "O", n_points)
        noise = oracle_noise(freq, den, bmag, bpsi, fine, "O", n_points, runs=8, seed=seed)
        for trim in (1, 0):
            library.set_option("trim_lds", trim)
            got = library.vertical_forward_operator(freq, den, bmag, bpsi, fine, "O", n_points)
            gotf = library.vertical_forward_operator(freq, den, bmag, bpsi, fine, "O", n_points, math=library.MATH_FAITHFUL)
            library.set_option("trim_lds", 1)
            err, ok = rel_err(got, want_o)
            errf, _ = rel_err(gotf, want_o)
            errc, _ = rel_err(want_c, want_o)
            over = ok & (err > 1e-6)
            if over.sum() > 1:
                print("iteration", it, "trim", trim, "levels", n_tall, "n_points", n_points, "pairs over 1e-6:", int(over.sum()), "of", int(ok.sum()))
                for idx in np.argwhere(over):
                    i, j = idx
                    print("  pair", idx, "f", freq[j], "err", err[i, j], "faithful-tier err", errf[i, j], "C-oracle vs NumPy", errc[i, j], "noise", noise[i, j], "vh", want_o[i, j])
