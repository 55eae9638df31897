import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_random as t
from oracle import vfo_numpy as orc
from pyrayhf_amd import library
np.seterr(all="ignore"); np.set_printoptions(precision=12, linewidth=200)
seed = 276
rng = np.random.default_rng(7000 + seed)
for it in range(60):
    freq, den, bmag, bpsi, alt, n_points = t.random_problem(rng)
    if np.any(np.argmax(den, axis=1) == 0): continue
    alt = np.array(alt, dtype=np.float64, copy=True)
    n_prof, n_alt = den.shape
    victim = int(rng.integers(n_prof))
    what = rng.choice(["den", "alt", "bmag", "bpsi", "bpsi", "bmag"])
    clean = bpsi.copy()
    if what == "den":
        first = int(rng.integers(1, n_alt)); den[victim, first:] = np.nan
    else:
        col = {"alt": alt if alt.ndim == 2 else None, "bmag": bmag, "bpsi": bpsi}[what]
        if col is None: alt[rng.integers(n_alt)] = np.nan
        else: col[victim, rng.integers(0, n_alt, int(rng.integers(1, 4)))] = np.nan
    if it == 8: break
p = 0; f = freq[3:4]; a = alt if alt.ndim == 1 else alt[p]
for name, ps in (("NaN", bpsi[p]), ("clean", clean[p])):
    for n in (2, 3, 64):
        w = orc.virtual_heights(f, den[p], bmag[p], ps, a, "O", n)
        g = library.vertical_forward_operator(f, den[p], bmag[p], ps, a, "O", n)
        gf = library.vertical_forward_operator(f, den[p], bmag[p], ps, a, "O", n, math=library.MATH_FAITHFUL)
        print(name, "n_points", n, "oracle", w, "gpu", g, "gpu faithful", gf, "rel", (g - w) / w)
