#!/usr/bin/env python3
"""The reference's stage functions on arrays of its own call sizes (one profile: 174 frequencies x 200 grid points), from
NumPy arrays and back: microseconds per call of regrid_to_nonuniform_grid, find_mu_mup, find_vh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrayhf_amd import library, synth

alt, den, bmag, bpsi = synth.chapman_profiles(1, 7)
den, bmag, bpsi = den[0], bmag[0], bpsi[0]
f = np.linspace(2e6, 0.98 * library.den2freq(den.max()), 174)

def timed(fn, n=200):
    for _ in range(10): r = fn()
    t = time.perf_counter()
    for _ in range(n): r = fn()
    return (time.perf_counter() - t) / n * 1e6, r

us, g = timed(lambda: library.regrid_to_nonuniform_grid(f, den, bmag, bpsi, alt, "O", 200))
print("regrid_to_nonuniform_grid us per call", round(us, 1))
X = library.find_X(g["den"], g["freq"]); Y = library.find_Y(g["freq"], g["bmag"]); ps = g["bpsi"]; dh = g["dist"]
us, r = timed(lambda: library.find_mu_mup(X, Y, ps, "O"))
print("find_mu_mup", X.shape, "us per call", round(us, 1))
us, r = timed(lambda: library.find_vh(X, Y, ps, dh, float(alt.min()), "O"))
print("find_vh", X.shape, "us per call", round(us, 1))
us, r = timed(lambda: library.vertical_forward_operator(f, den, bmag, bpsi, alt, "O", 200))
print("vertical_forward_operator us per call", round(us, 1))
