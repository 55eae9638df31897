#!/usr/bin/env python3
"""Device-resident timing of the un-fused stage ops: these ARE HBM-bound, so each line
reports algorithmic bytes / kernel time against the 8 TB/s roofline.  One JSON object per op."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrayhf_amd import _native, library, synth

dev = torch.device("cuda", 0)
ctx = _native.context(0)
DP = _native.FLAG_DEVICE_PTRS
HBM = 8000.0


def report(name, bytes_moved, ms, extra=None):
    gbs = bytes_moved / (ms * 1e-3) / 1e9
    rec = {"op": name, "kernel_ms": ms, "algorithmic_bytes": bytes_moved,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM, "unit": "GB/s", "frac": gbs / HBM}}
    rec.update(extra or {})
    print(json.dumps(rec), flush=True)


def best(fn, reps=5):
    ms = []
    for _ in range(reps):
        _native.raise_for(fn())
        ms.append(ctx.last_kernel_ms())
    return min(ms[1:])


rng = np.random.default_rng(0)
# --- find_mu_mup on a flat field: 24 B in, 16 B out per element
n = 1 << 26
X = torch.rand(n, dtype=torch.float64, device=dev) * 0.9
Y = torch.rand(n, dtype=torch.float64, device=dev) * 0.3 + 0.05
P = torch.rand(n, dtype=torch.float64, device=dev) * 90.0
mu = torch.empty_like(X); mup = torch.empty_like(X)
for tier, label in ((_native.MATH_FAITHFUL, "faithful"), (_native.MATH_FAST, "fast")):
    ctx.set_math(tier)
    ms = best(lambda: ctx.mu_mup(X.data_ptr(), Y.data_ptr(), P.data_ptr(), n, 1, mu.data_ptr(), mup.data_ptr(), DP))
    report(f"find_mu_mup X-mode, {n} elements, {label} (one pass: the nanmax|Y| test rides along)", 40 * n, ms,
           {"elements_per_s": n / (ms * 1e-3)})

# --- find_vh: (F, N) arrays X, Y, psi, dh -> (F,)
F, N = 4096, 20000
Xr = torch.rand(F, N, dtype=torch.float64, device=dev) * 0.9
Yr = torch.rand(F, N, dtype=torch.float64, device=dev) * 0.3 + 0.05
Pr = torch.rand(F, N, dtype=torch.float64, device=dev) * 90.0
Dr = torch.rand(F, N, dtype=torch.float64, device=dev) * 0.01
vh = torch.empty(F, dtype=torch.float64, device=dev)
for tier, label in ((_native.MATH_FAITHFUL, "faithful"), (_native.MATH_FAST, "fast")):
    ctx.set_math(tier)
    ms = best(lambda: ctx.find_vh(Xr.data_ptr(), Yr.data_ptr(), Pr.data_ptr(), Dr.data_ptr(), F, N, 80.0, 1,
                                  vh.data_ptr(), DP))
    report(f"find_vh X-mode, ({F}, {N}), {label} (one pass: the nanmax|Y| test rides along)", 32 * F * N, ms,
           {"points_per_s": F * N / (ms * 1e-3)})
del Xr, Yr, Pr, Dr

# --- regrid_to_nonuniform_grid: one profile, F frequencies -> eight (F, N) arrays, 64 B written per point
alt, den, bmag, bpsi = synth.chapman_profiles(1, 5)
F, N = 1024, 20000
freq_hz = torch.as_tensor(np.linspace(0.5, 16.0, F) * 1e6, device=dev)
t = [torch.as_tensor(x, device=dev) for x in (den[0], bmag[0], bpsi[0], alt)]
mult = torch.as_tensor(library._multiplier(N), device=dev)
outs = [torch.empty(F, N, dtype=torch.float64, device=dev) for _ in range(7)] + \
       [torch.empty(F, N, dtype=torch.int64, device=dev)]
ptrs = [o.data_ptr() for o in outs]
ms = best(lambda: ctx.regrid(freq_hz.data_ptr(), F, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                             alt.size, mult.data_ptr(), N, 1, ptrs, DP))
report(f"regrid_to_nonuniform_grid X-mode, 1 profile x {F} freqs x {N} points", 64 * F * N, ms,
       {"points_per_s": F * N / (ms * 1e-3)})
