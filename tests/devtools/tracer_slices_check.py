#!/usr/bin/env python3
"""The per-ray tracer launch cuts its rays into eight slices with a queue each (prhf_snell.inc snell_dispatch): ray
counts that are not multiples of anything, from 1 to 2 000 003, must give every ray the result it has in a launch of its
own size - compared with the same rays traced in pieces of 1 777.  In the reference's operation order: there the
per-profile table (taken when the rays outnumber the profiles four to one) changes no bit, so every size must agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyrayhf_amd import library, synth, tracers

alt, den, bmag, bpsi = synth.chapman_profiles(64, 3)
rng = np.random.default_rng(9)
R = 2000003
f = rng.uniform(2e6, 14e6, R); e = rng.uniform(1.0, 90.0, R); idx = rng.integers(0, 64, R)
bad = 0
for name, fn in (("flat", tracers.trace_rays_cartesian_snells), ("spherical", tracers.trace_rays_spherical_snells)):
    whole = fn(f, e, alt, den, bmag, bpsi, "O", profile_index=idx, math=library.MATH_FAITHFUL)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 255, 257, 1777, 40961):
        part = fn(f[:n], e[:n], alt, den, bmag, bpsi, "O", profile_index=idx[:n], math=library.MATH_FAITHFUL)
        for k in ("group_path_km", "group_delay_sec", "ground_range_km", "x_midpoint", "z_midpoint", "n_path"):
            if not np.array_equal(part[k], whole[k][:n], equal_nan=True):
                bad += 1; print("MISMATCH", name, n, k)
    for start in range(0, 200000, 1777):
        part = fn(f[start:start + 1777], e[start:start + 1777], alt, den, bmag, bpsi, "O", profile_index=idx[start:start + 1777], math=library.MATH_FAITHFUL)
        if not np.array_equal(part["group_path_km"], whole["group_path_km"][start:start + 1777], equal_nan=True):
            bad += 1; print("MISMATCH", name, "piece at", start)
    print(name, "rays", R, "turned", int(np.isfinite(whole["group_path_km"]).sum()), "every output written:",
          bool((whole["n_path"] >= 0).all()))
print("mismatches", bad)
sys.exit(1 if bad else 0)
