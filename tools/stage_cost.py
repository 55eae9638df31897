"""Cost split of the fused kernel on the config-4 shard: staging only (one escaping frequency),
staging + 256 reflection searches (all frequencies escape), and the full sweep."""
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
run("stage only: 1 escaping frequency", [30.0])
run("stage + 256 escaping frequencies", np.full(256, 30.0))
run("stage + 256 escaping, faithful tier staging", np.full(256, 30.0), math=0)
run("stage + 64 escaping frequencies", np.full(64, 30.0))
run("full config-4 sweep", synth.sounder_frequencies(4))
run("full sweep, n_points 2000", synth.sounder_frequencies(4), n_points=2000)
run("full sweep, n_points 200", synth.sounder_frequencies(4), n_points=200)
