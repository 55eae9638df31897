#!/usr/bin/env python3
"""Workload for profiling the Snell's-law tracer kernels (PROF_CMD of tools/profile.sh): 200 000 rays with random
frequency, elevation and profile, flat then spherical Earth, three launches each.  Prints kernel times."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrayhf_amd import tracers, synth, _native
import _options
opts = _options.apply()                 # PRHF_TOOL_OPTIONS="name=value,..."

alt, den, bmag, bpsi = synth.chapman_profiles(256, 7)
rng = np.random.default_rng(0)
R = 200000
f = rng.uniform(2e6, 14e6, R); e = rng.uniform(5.0, 89.0, R); idx = rng.integers(0, 256, R)
if os.environ.get("PRHF_SORT_RAYS"):      # rays in profile order (what binning them before the launch would buy)
    idx = np.sort(idx)
ctx = _native.context(0)
for name, fn in (("cartesian", tracers.trace_rays_cartesian_snells), ("spherical", tracers.trace_rays_spherical_snells)):
    for rep in range(3):
        r = fn(f, e, alt, den, bmag, bpsi, "O", profile_index=idx)
    kms = ctx.last_kernel_ms()
    print(json.dumps({"tracer": name, "rays": R, "traced_fraction": float(np.isfinite(r["group_path_km"]).mean()),
                      "options": opts, "kernel_ms": kms, "rays_per_s_kernel": R / (kms * 1e-3)}), flush=True)
