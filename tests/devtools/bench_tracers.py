#!/usr/bin/env python3
"""Rays per second of the Snell's-law tracers (device time from HIP events) beside the oracle on one
host core.  The reference quotes 1.4 ms (flat) and 2.3 ms (spherical) per ray on an unspecified laptop
(docs/tutorials/Example_PyRayHF_{Cartesian,Spherical}_Snells.ipynb, cell 1 outputs)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyrayhf_amd import tracers, synth, _native
from oracle import snell_numpy as sn

alt, den, bmag, bpsi = synth.chapman_profiles(256, 7)
rng = np.random.default_rng(0)
R = 200000
f = rng.uniform(2e6, 14e6, R); e = rng.uniform(5.0, 89.0, R); idx = rng.integers(0, 256, R)
ctx = _native.context(0)
for name, fn, ofn in (("cartesian", tracers.trace_rays_cartesian_snells, sn.trace_cartesian),
                      ("spherical", tracers.trace_rays_spherical_snells, sn.trace_spherical)):
    for rep in range(3):
        t0 = time.perf_counter(); r = fn(f, e, alt, den, bmag, bpsi, "O", profile_index=idx); wall = time.perf_counter() - t0
        kms = ctx.last_kernel_ms()
    n_cpu = 200
    t0 = time.perf_counter()
    for k in range(n_cpu):
        ofn(f[k], e[k], alt, den[idx[k]], bmag[idx[k]], bpsi[idx[k]], "O")
    cpu = (time.perf_counter() - t0) / n_cpu
    print(json.dumps({"tracer": name, "rays": R, "traced_fraction": float(np.isfinite(r["group_path_km"]).mean()),
                      "kernel_ms": kms, "rays_per_s_kernel": R / (kms * 1e-3), "wall_ms_host_buffers": 1e3 * wall,
                      "oracle_ms_per_ray_1core": 1e3 * cpu, "oracle_rays_per_s": 1.0 / cpu}), flush=True)
