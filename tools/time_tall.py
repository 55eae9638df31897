#!/usr/bin/env python3
"""Cost of profiles that do not fit LDS (vfo_tall_kernel: staged in global memory, generic loop): the same Chapman
layers sampled on 1 400 levels (LDS, main loop), 1 401 levels (tall) and 6 200 levels (tall, 0.1 km)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0); ctx = _native.context(0)
P = 1000
a0, den0, bmag0, bpsi0 = synth.chapman_profiles(P, 20260004)
freq = synth.sounder_frequencies(4)

def run(name, n_alt, mode, n):
    alt = np.linspace(80.0, 699.0, n_alt)
    cols = [np.stack([np.interp(alt, a0, r) for r in x]) for x in (den0, bmag0, bpsi0)]
    t = [torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (freq, *cols, alt)]
    ms = []
    for _ in range(3):
        out = library.vertical_forward_operator(*t, mode, n); ms.append(ctx.last_kernel_ms())
    fin = float(np.isfinite(out.cpu().numpy()).mean())
    print(json.dumps({"case": name, "n_alt": n_alt, "mode": mode, "n_points": n, "profiles": P, "freqs": int(freq.size),
                      "kernel_ms": min(ms[1:]), "finite": fin,
                      "ns_per_finite_pair_and_1000_points": min(ms[1:]) * 1e6 / (P * freq.size * fin * n / 1000.0)}), flush=True)

for mode, n in (("X", 20000), ("X", 2000), ("O", 200)):
    run("LDS (1400 levels)", 1400, mode, n)
    run("1401 levels, bottomsides fit LDS (staged up to the highest peak)", 1401, mode, n)
    run("2480 levels (0.25 km), bottomsides fit LDS", 2480, mode, n)
    library.set_option("trim_lds", 0)
    run("1401 levels, global-memory slabs (trim_lds = 0)", 1401, mode, n)
    library.set_option("trim_lds", 1)
    run("6200 levels (0.1 km): global-memory slabs", 6200, mode, n)
# ... and the generic loop on the same slabs (what tall profiles took up to round 4): option tall_lean = 0
library.set_option("tall_lean", 0)
for mode, n in (("X", 20000), ("X", 2000), ("O", 200)):
    library.set_option("trim_lds", 0)
    run("1401 levels, global-memory slabs, generic loop (tall_lean = 0)", 1401, mode, n)
    library.set_option("trim_lds", 1)
    run("6200 levels (0.1 km): global-memory slabs, generic loop (tall_lean = 0)", 6200, mode, n)
library.set_option("tall_lean", 1)
