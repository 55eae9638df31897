import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_gpu_random import random_problem
from parity import rel_err
from oracle import vfo_c, vfo_numpy
from pyrayhf_amd import library
rng = np.random.default_rng(1001)
for it in range(40):
    freq, den, bmag, bpsi, alt, n_points = random_problem(rng)
    if np.any(np.argmax(den, axis=1) == 0): continue
    want = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, "O", n_points)
    wnp = vfo_numpy.virtual_heights_batch(freq, den, bmag, bpsi, alt, "O", n_points)
    for tier in (0, 1):
        got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", n_points, math=tier)
        err, ok = rel_err(got, want)
        e2, _ = rel_err(wnp, want)
        if ok.sum() >= 10 and np.mean(err[ok] <= 1e-6) < 0.9:
            print(it, "tier", tier, "n_points", n_points, "n_alt", den.shape[1], "ok", ok.sum(), "frac", np.mean(err[ok] <= 1e-6),
                  "max", err.max(), "median", np.median(err[ok]), "| numpy-vs-C frac", np.mean(e2[ok] <= 1e-6), "max", e2.max())
