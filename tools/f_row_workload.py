#!/usr/bin/env python3
"""The tracer launches of bench.py's f-row legs, three times each in a fixed order (per-ray flat, per-ray spherical, fan
flat, fan spherical), for tools/f_row_profile.sh: the profiler's per-dispatch counters then give the vector instructions
per ray that bench.py prices each leg against (profiles/f_row_bounds.json)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrayhf_amd import tracers, synth, _native
ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(256, 7)
rng = np.random.default_rng(0)
R = 200000
f = rng.uniform(2e6, 14e6, R); e = rng.uniform(5.0, 89.0, R); idx = rng.integers(0, 256, R)
fan_f, fan_e = np.linspace(2e6, 14e6, 100), np.linspace(5.0, 89.0, 128)
for name, fn in (("snell_per_ray_flat", lambda: tracers.trace_rays_cartesian_snells(f, e, alt, den, bmag, bpsi, "O", profile_index=idx)),
                 ("snell_per_ray_spherical", lambda: tracers.trace_rays_spherical_snells(f, e, alt, den, bmag, bpsi, "O", profile_index=idx)),
                 ("snell_fan_flat", lambda: tracers.trace_fan_cartesian_snells(fan_f, fan_e, alt, den[:16], bmag[:16], bpsi[:16], "O")),
                 ("snell_fan_spherical", lambda: tracers.trace_fan_spherical_snells(fan_f, fan_e, alt, den[:16], bmag[:16], bpsi[:16], "O"))):
    ms = []
    for _ in range(3):
        r = fn()
        ms.append(ctx.last_kernel_ms())
    n = r["group_path_km"].size
    print(json.dumps({"leg": name, "rays": int(n), "kernel_ms": min(ms[1:]), "rays_per_s": n / (min(ms[1:]) * 1e-3)}), flush=True)
