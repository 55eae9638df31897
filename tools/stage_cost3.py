"""Cost split of config 3 (10 000 profiles x 174 freqs, O mode, n_points = 200)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0); ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t = [torch.as_tensor(x, device=dev) for x in (den, bmag, bpsi, alt)]
def run(name, freq, n_points=200, math=None):
    f = torch.as_tensor(np.asarray(freq, dtype=np.float64), device=dev)
    ms = []
    for r in range(4):
        out = library.vertical_forward_operator(f, *t, "O", n_points, math=math); ms.append(ctx.last_kernel_ms())
    print(json.dumps({"case": name, "kernel_ms": min(ms[1:]), "finite": float(np.isfinite(out.cpu().numpy()).mean())}), flush=True)
f174 = synth.sounder_frequencies(3)
run("staging only: 1 escaping frequency", [30.0])
run("174 certainly escaping frequencies", np.full(174, 30.0))
run("config 3", f174)
run("config 3, n_points 2", f174, n_points=2)
run("config 3, n_points 64", f174, n_points=64)
run("config 3, n_points 128", f174, n_points=128)
run("config 3, n_points 400", f174, n_points=400)
