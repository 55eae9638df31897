#!/usr/bin/env python3
"""Kernel time of every BASELINE.json configuration that fits one GPU (device-resident inputs).
Prints one JSON object per configuration; numbers for DESIGN.md section 6."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrayhf_amd import library, synth, _native

dev = torch.device("cuda", 0)
ctx = _native.context(0)

def run(name, freq, alt, den, bmag, bpsi, mode, n_points, reps=5, math=None):
    t = [torch.as_tensor(x, device=dev) for x in (freq, den, bmag, bpsi, alt)]
    out = None
    ms = []
    for r in range(reps + 2):
        out = library.vertical_forward_operator(*t, mode, n_points, math=math, sync=True)
        if r >= 2:
            ms.append(ctx.last_kernel_ms())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        library.vertical_forward_operator(*t, mode, n_points, math=math, sync=False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    pairs = (den.shape[0] if den.ndim == 2 else 1) * freq.size
    print(json.dumps({"config": name, "mode": mode, "n_points": n_points, "pairs": pairs, "math": math,
                      "kernel_ms": float(np.median(ms)), "wall_ms_per_call": 1e3 * wall,
                      "integrals_per_s_kernel": pairs / (np.median(ms) * 1e-3),
                      "reflecting": float(np.isfinite(out.cpu().numpy()).mean())}), flush=True)

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g4_day_night.npz"))
day = [g["Day_" + k] for k in ("alt", "den", "bmag", "bpsi")]
f174 = synth.sounder_frequencies(1)
run("1: Day profile x 174, O/200", f174, *day, "O", 200, reps=20)
run("2: Day profile x 174, X/20000", f174, *day, "X", 20000, reps=20)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
run("3: 10000 Chapman x 174, O/200", f174, alt, den, bmag, bpsi, "O", 200)
run("3 (fast tier): 10000 Chapman x 174, O/200", f174, alt, den, bmag, bpsi, "O", 200, math=library.MATH_FAST)
run("3 (faithful everywhere): 10000 Chapman x 174, O/200", f174, alt, den, bmag, bpsi, "O", 200, math=library.MATH_FAITHFUL)
run("3b: 2000 Chapman x 174, O/20000", f174, alt, den[:2000], bmag[:2000], bpsi[:2000], "O", 20000, reps=3)
run("3b (faithful everywhere): 2000 Chapman x 174, O/20000", f174, alt, den[:2000], bmag[:2000], bpsi[:2000], "O", 20000, reps=2, math=library.MATH_FAITHFUL)
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, 12500))
run("4 shard: 12500 Chapman x 256, X/20000", synth.sounder_frequencies(4), alt, den, bmag, bpsi, "X", 20000, reps=3)
run("4 shard (faithful): 12500 Chapman x 256, X/20000", synth.sounder_frequencies(4), alt, den, bmag, bpsi, "X", 20000,
    reps=2, math=library.MATH_FAITHFUL)
# config 5, per-GPU shard: rank 0 of 8 takes its block of every slice (dist.shard_segments), one launch
from pyrayhf_amd import dist as pdist
CONFIG5 = [(0, 20000, "O", 200), (20000, 35000, "X", 2000), (35000, 45000, "O", 2000), (45000, 50000, "X", 20000)]
rows, segs = pdist.shard_segments(CONFIG5, 8, 0)
alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=rows)
P = den.shape[0]
f512 = synth.sounder_frequencies(5)
tt = [torch.as_tensor(x, device=dev) for x in (f512, den, bmag, bpsi, alt)]
for r in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    library.vertical_forward_operator_mixed(*tt, segs)
    wall = time.perf_counter() - t0
    kms = ctx.last_kernel_ms()
print(json.dumps({"config": "5 shard: 6250 Chapman x 512, mixed O/X x {200,2000,20000}, one launch", "pairs": P * 512,
                  "kernel_ms": kms, "wall_ms_per_call": 1e3 * wall,
                  "integrals_per_s_kernel": P * 512 / (kms * 1e-3)}), flush=True)
