#!/usr/bin/env python3
"""How much slower are the less regular inputs: a non-uniform altitude grid (main loop with a monotone segment cursor),
a field angle that turns too fast for the per-segment polynomial (main loop with the rotation form) and one that turns
by more than 0.05 rad per level (sin per point, generic loop)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0); ctx = _native.context(0)
P = 2000
rng = np.random.default_rng(11)
alt_u, den, bmag, bpsi = synth.chapman_profiles(P, 20260004)
freq = synth.sounder_frequencies(4)

def chap(alt):
    r = np.random.default_rng(20260004)
    nmf2 = 10.0 ** r.uniform(11.3, 12.5, size=P); hmf2 = r.uniform(220.0, 420.0, size=P); hf2 = r.uniform(35.0, 70.0, size=P)
    nme = 10.0 ** r.uniform(10.3, 11.3, size=P); he = r.uniform(6.0, 12.0, size=P); b0 = r.uniform(2.2e-5, 6.0e-5, size=P)
    psi0 = r.uniform(0.0, 89.0, size=P)
    c = lambda nm, hm, h: nm[:, None] * np.exp(0.5 * (1 - (alt[None] - hm[:, None]) / h[:, None] - np.exp(-(alt[None] - hm[:, None]) / h[:, None])))
    return (c(nmf2, hmf2, hf2) + c(nme, np.full(P, 110.0), he), b0[:, None] * ((6371.0 + 80.0) / (6371.0 + alt[None])) ** 3,
            psi0[:, None] + 0.001 * (alt[None] - 80.0))

def run(name, alt, den, bmag, bpsi, mode="X", n=20000):
    t = [torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (freq, den, bmag, bpsi, alt)]
    ms = []
    for _ in range(3):
        out = library.vertical_forward_operator(*t, mode, n); ms.append(ctx.last_kernel_ms())
    print(json.dumps({"case": name, "kernel_ms": min(ms[1:]), "finite": float(np.isfinite(out.cpu().numpy()).mean())}), flush=True)

run("uniform grid, slowly turning field (main loop)", alt_u, den, bmag, bpsi)
alt_n = 80.0 + np.concatenate([[0.0], np.cumsum(rng.uniform(0.5, 1.5, alt_u.size - 1))])
run("non-uniform grid (main loop, segment cursor)", alt_n, *chap(alt_n))
d2, b2, p2 = chap(alt_u)
run("field angle turning 0.05 deg/km (main loop, rotation form)", alt_u, d2, b2, p2 + 0.05 * (alt_u[None] - 80.0))
run("field angle turning 1 deg/km (main loop, rotation form)", alt_u, d2, b2, p2 + 1.0 * (alt_u[None] - 80.0))
run("field angle turning 4 deg/km (sin per point, generic loop)", alt_u, d2, b2, p2 + 4.0 * (alt_u[None] - 80.0))
run("non-uniform grid and a field angle turning 0.05 deg/km", alt_n, *[x if i < 2 else x + 0.05 * (alt_n[None] - 80.0) for i, x in enumerate(chap(alt_n))])
run("O mode default, uniform grid", alt_u, den, bmag, bpsi, mode="O")
run("O mode default, non-uniform grid", alt_n, *chap(alt_n), mode="O")
