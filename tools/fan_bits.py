#!/usr/bin/env python3
"""Outputs of the grouped tracer launches on a fixed set of fans, saved for a bit-for-bit comparison between two builds
of the library (PRHF_LIB): tools/fan_bits.py out.npz, then tools/fan_bits.py --compare a.npz b.npz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

KEYS = ("group_path_km", "group_delay_sec", "ground_range_km", "x_turn_km", "z_turn_km", "x_midpoint", "z_midpoint",
        "n_path", "x", "z")
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    for k in bad:
        d = np.abs(a[k] - b[k]); w = np.nanmax(d / np.maximum(np.abs(b[k]), 1e-300))
        print(k, "differs: worst relative", w, "nan pattern equal", np.array_equal(np.isnan(a[k]), np.isnan(b[k])))
    print("arrays", len(a.files), "different", len(bad))
    sys.exit(1 if bad else 0)
from pyrayhf_amd import tracers, synth
out = {}
alt, den, bmag, bpsi = synth.chapman_profiles(12, 11)
bmag[2] = 0.0
den2 = den.copy(); den2[5, 100:140] = 0.0            # (levels of vacuum in the column)
alt_nan = alt.copy(); alt_nan[..., 150] = np.nan    # (a criterion mu r that is NaN at one entry: the bracket walk's fallback)
f = np.linspace(1.5e6, 15e6, 40); e = np.concatenate([np.linspace(0.0, 90.0, 46), [89.99, 90.5, -3.0]])
for mode in "OX":
    for name, fn, kw in (("flat", tracers.trace_fan_cartesian_snells, {}), ("sph", tracers.trace_fan_spherical_snells, {}),
                         ("sph_small", tracers.trace_fan_spherical_snells, {"R_E": 1000.0, "dz_target_km": 0.3}),
                         ("sph_nan_alt", tracers.trace_fan_spherical_snells, {"alt": alt_nan})):
        kw = dict(kw)
        grid = kw.pop("alt", alt)
        for tag, d in (("a", den), ("b", den2)):
            r = fn(f, e, grid, d, bmag, bpsi, mode, return_paths=True, **kw)
            for k in KEYS:
                out[f"{name}_{mode}_{tag}_{k}"] = r[k]
np.savez(sys.argv[1], **out)
print("saved", len(out), "arrays; rays that turn:", {k: int(np.isfinite(v).sum()) for k, v in out.items() if k.endswith("a_group_path_km")})
