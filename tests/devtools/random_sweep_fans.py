#!/usr/bin/env python3
"""Fresh-seed sweep of the grouped tracer launch (prhf_snell_fan_f64) against the per-ray launch in the reference's
operation order, which finds its brackets by walking the levels: random Chapman columns with a random second layer, a
band of vacuum, an unmagnetised column, on a spherical Earth sometimes a NaN altitude; random frequencies and
elevations (some outside [0, 90]); both modes.  Usage: random_sweep_fans.py first_seed count"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyrayhf_amd import library, synth, tracers

KEYS = ("group_path_km", "group_delay_sec", "ground_range_km", "x_turn_km", "z_turn_km", "x_midpoint", "z_midpoint", "x", "z")
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = rays_total = turned = 0
worst = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    P = int(rng.integers(1, 7))
    alt, den, bmag, bpsi = synth.chapman_profiles(P, seed)
    den = den.copy(); bmag = bmag.copy()
    for p in range(P):
        kind = rng.integers(0, 4)
        if kind == 0:
            den[p] += rng.uniform(0.2, 3.0) * den[p].max() * np.exp(-0.5 * ((alt - rng.uniform(95, 130)) / rng.uniform(3, 10)) ** 2)
        elif kind == 1:
            k0 = int(rng.integers(20, 300)); den[p, k0:k0 + int(rng.integers(1, 60))] = 0.0
        elif kind == 2:
            bmag[p] = 0.0
    spherical = bool(rng.integers(0, 2))
    grid = alt
    if spherical and rng.random() < 0.3:
        grid = alt.copy(); grid[int(rng.integers(5, 300))] = np.nan
    F, E = int(rng.integers(1, 12)), int(rng.integers(1, 40))
    freqs = rng.uniform(1.2e6, 16e6, F)
    elevs = np.concatenate([rng.uniform(0.0, 90.0, E), rng.choice([0.0, 90.0, 89.999, 90.5, -2.0], 2)])
    kw = {"dz_target_km": float(rng.choice([0.3, 1.0, 3.0])), "max_substeps": int(rng.choice([50, 400]))} if spherical else {}
    fan_fn = tracers.trace_fan_spherical_snells if spherical else tracers.trace_fan_cartesian_snells
    ray_fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
    pp, ff, ee = np.meshgrid(np.arange(P), freqs, elevs, indexing="ij")
    for mode in "OX":
        fan = fan_fn(freqs, elevs, grid, den, bmag, bpsi, mode, return_paths=True, **kw)
        rays = ray_fn(ff.ravel(), ee.ravel(), grid, den, bmag, bpsi, mode, profile_index=pp.ravel(), return_paths=True,
                      math=library.MATH_FAITHFUL, **kw)
        rays_total += ff.size
        turned += int(np.isfinite(rays["group_path_km"]).sum())
        ok = np.array_equal(fan["n_path"].ravel(), rays["n_path"])
        for key in KEYS:
            got, want = fan[key].reshape(rays[key].shape), rays[key]
            if not np.array_equal(np.isnan(got), np.isnan(want)):
                ok = False
                continue
            with np.errstate(invalid="ignore", divide="ignore"):
                d = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
            w = float(np.nanmax(d)) if np.isfinite(d).any() else 0.0
            worst = max(worst, w)
            if w > 1e-13:
                ok = False
        if not ok:
            bad += 1
            print("VIOLATION seed", seed, "mode", mode, "spherical", spherical, flush=True)
print(f"seeds {first}..{first + count - 1}: {rays_total} rays ({turned} turn), worst relative difference {worst:.2e}, violations {bad}")
sys.exit(1 if bad else 0)
