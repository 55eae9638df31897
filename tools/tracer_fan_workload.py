#!/usr/bin/env python3
"""Fans of rays (oblique-ionogram shape): P profiles x F frequencies x E elevations, per-ray launch against the grouped
launch that computes the refractive-index levels once per (profile, frequency)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrayhf_amd import tracers, synth, _native
import _options
opts = _options.apply()                 # PRHF_TOOL_OPTIONS="name=value,..."

P, F, E = 16, 100, 128
alt, den, bmag, bpsi = synth.chapman_profiles(P, 7)
f = np.linspace(2e6, 14e6, F); e = np.linspace(5.0, 89.0, E)
ctx = _native.context(0)
pp, ff, ee = np.meshgrid(np.arange(P), f, e, indexing="ij")
for name, fan_fn, ray_fn in (("cartesian", tracers.trace_fan_cartesian_snells, tracers.trace_rays_cartesian_snells),
                             ("spherical", tracers.trace_fan_spherical_snells, tracers.trace_rays_spherical_snells)):
    for rep in range(3):
        a = fan_fn(f, e, alt, den, bmag, bpsi, "O")
    fan_ms = ctx.last_kernel_ms()
    for rep in range(3):
        b = ray_fn(ff.ravel(), ee.ravel(), alt, den, bmag, bpsi, "O", profile_index=pp.ravel())
    ray_ms = ctx.last_kernel_ms()
    pa, pb = a["group_path_km"].ravel(), b["group_path_km"]
    same = bool(np.array_equal(np.isnan(pa), np.isnan(pb)))
    worst = float(np.nanmax(np.abs(pa - pb) / np.abs(pb))) if np.isfinite(pb).any() else 0.0
    print(json.dumps({"tracer": name, "rays": int(ff.size), "groups": P * F, "fan_kernel_ms": fan_ms, "per_ray_kernel_ms": ray_ms,
                      "rays_per_s_fan": ff.size / (fan_ms * 1e-3), "rays_per_s_per_ray": ff.size / (ray_ms * 1e-3),
                      "same_rays_turn": same, "worst_path_difference": worst, "traced_fraction": float(np.isfinite(b["group_path_km"]).mean())}), flush=True)
