"""The four slices of the config-5 per-GPU shard timed alone (their own launches, default arithmetic) beside the
mixed work-list launch: what mixing costs, and which slice carries the time."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from pyrayhf_amd import library, synth, dist as pdist, _native
dev = torch.device("cuda", 0)
ctx = _native.context(0)
segs = bench.config5_segments(1)
rows, local = pdist.shard_segments(segs, 1, 0)
alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=rows)
freq = synth.sounder_frequencies(5)
t = [torch.as_tensor(x, device=dev) for x in (freq, den, bmag, bpsi, alt)]
def best(fn):
    ms = []
    for _ in range(4):
        fn(); ms.append(ctx.last_kernel_ms())
    return min(ms[1:])
total = 0.0
for (p0, p1, mode, n) in local:
    ms = best(lambda: library.vertical_forward_operator(t[0], t[1][p0:p1], t[2][p0:p1], t[3][p0:p1], t[4], mode, n))
    out = library.vertical_forward_operator(t[0], t[1][p0:p1], t[2][p0:p1], t[3][p0:p1], t[4], mode, n)
    fin = float(torch.isfinite(out).double().mean())
    total += ms
    print(json.dumps({"slice": f"{p1 - p0} profiles x {freq.size} freqs, {mode}/{n}", "kernel_ms": ms, "finite": fin,
                      "ns_per_finite_pair_and_1000_points": 1e6 * ms / (fin * (p1 - p0) * freq.size * n / 1000.0)}), flush=True)
mixed = best(lambda: library.vertical_forward_operator_mixed(*t, local))
print(json.dumps({"sum_of_slices_ms": total, "mixed_launch_ms": mixed}))
