"""Cost split of the fused kernel on the config-4 shard (12 500 profiles x 256 freqs, X mode): staging only,
escaping pairs, and the full sweep at several grid sizes - the slope over n_points is the cost per grid
point, the intercept the fixed cost per pair (scan, set-up, reduction)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0)
ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, 12500))
t = [torch.as_tensor(x, device=dev) for x in (den, bmag, bpsi, alt)]
def run(name, freq, mode="X", n_points=20000, math=None):
    f = torch.as_tensor(np.asarray(freq, dtype=np.float64), device=dev)
    ms = []
    for r in range(4):
        library.vertical_forward_operator(f, *t, mode, n_points, math=math)
        ms.append(ctx.last_kernel_ms())
    print(json.dumps({"case": name, "kernel_ms": min(ms[1:])}), flush=True)
    return min(ms[1:])
f4 = synth.sounder_frequencies(4)
run("stage only: 1 escaping frequency", [30.0])
run("stage + 256 certainly escaping frequencies", np.full(256, 30.0))
times = {n: run(f"full sweep, n_points {n}", f4, n_points=n) for n in (320, 1280, 5120, 10000, 20000, 40000)}
slope = (times[40000] - times[10000]) / 30000
print(json.dumps({"ms_per_grid_point_of_all_pairs": slope, "intercept_ms": times[20000] - 20000 * slope,
                  "note": "time(n) ~ intercept + slope * n over n = 10000..40000"}))
