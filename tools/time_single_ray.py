import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pyrayhf_amd import tracers, synth
alt, den, bmag, bpsi = synth.chapman_profiles(1, 7)
den, bmag, bpsi = den[0], bmag[0], bpsi[0]
for name, fn in (("cartesian", tracers.trace_ray_cartesian_snells), ("spherical", tracers.trace_ray_spherical_snells)):
    for _ in range(20): r = fn(6e6, 40.0, alt, den, bmag, bpsi, "O")
    t = time.perf_counter()
    for _ in range(300): r = fn(6e6, 40.0, alt, den, bmag, bpsi, "O")
    dt = (time.perf_counter() - t) / 300
    print(name, "us per single-ray call", round(dt * 1e6, 1), "path nodes", len(r["x"]), "group path", r["group_path_km"])
