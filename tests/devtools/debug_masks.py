"""Find the random problems whose NaN mask differs from the C oracle and print the offending pairs."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_gpu_random import random_problem
from oracle import vfo_c
from pyrayhf_amd import library
for seed in range(6):
    rng = np.random.default_rng(1000 + seed)
    for it in range(40):
        freq, den, bmag, bpsi, alt, n_points = random_problem(rng)
        if np.any(np.argmax(den, axis=1) == 0):
            continue
        for mode in "OX":
            want = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, mode, n_points)
            for tier in (0, 1):
                got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points, math=tier)
                bad = np.isnan(got) != np.isnan(want)
                if bad.any():
                    for (p, f) in np.argwhere(bad)[:4]:
                        a = alt if alt.ndim == 1 else alt[p]
                        K = int(np.argmax(den[p]))
                        print(f"seed {seed} it {it} mode {mode} tier {tier} n_points {n_points} pair ({p},{f}) freq {freq[f]:.6f} "
                              f"got {got[p, f]!r} want {want[p, f]!r} K {K} n_alt {a.size} uniform {np.allclose(np.diff(a), a[1]-a[0])} "
                              f"den0 {den[p,0]:.3e} den1 {den[p,1]:.3e} b0 {bmag[p,0]:.3e} psi0 {bpsi[p,0]:.3f} fH {2.799249247e10*bmag[p,0]/1e6:.4f} "
                              f"fN0 {8.97866275*np.sqrt(den[p,0])/1e6:.4e}", flush=True)
print("done")
