"""Cost split of config 3 (10 000 profiles x 174 freqs, O mode, n_points = 200): staging, level scan,
integration by arithmetic setting and grid size.  PRHF_TOOL_OPTIONS="name=value,..." sets context options (tools/_options.py)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0); ctx = _native.context(0)
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import _options
OPTIONS = _options.apply()
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t = [torch.as_tensor(x, device=dev) for x in (den, bmag, bpsi, alt)]
def run(name, freq, n_points=200, math=None, mode="O"):
    f = torch.as_tensor(np.asarray(freq, dtype=np.float64), device=dev)
    ms = []
    for r in range(5):
        out = library.vertical_forward_operator(f, *t, mode, n_points, math=math); ms.append(ctx.last_kernel_ms())
    print(json.dumps({"case": name, "kernel_ms": min(ms[1:]), "finite": float(np.isfinite(out.cpu().numpy()).mean()),
                      "options": OPTIONS}), flush=True)
f174 = synth.sounder_frequencies(3)
run("staging only: 1 escaping frequency", [30.0])
run("174 certainly escaping frequencies (no scan)", np.full(174, 30.0))
run("174 frequencies that scan every level and escape by a hair", np.full(174, 17.45))
run("config 3 (default arithmetic)", f174)
run("config 3, reference order everywhere", f174, math=library.MATH_FAITHFUL)
run("config 3, reduced algebra everywhere", f174, math=library.MATH_FAST)
run("config 3 shape in X mode (fast tier)", f174, mode="X")
for n in (2, 64, 128, 136, 192, 256, 400, 2000):
    run(f"config 3, n_points {n}", f174, n_points=n)
run("config 3, n_points 2000, reduced algebra everywhere", f174, n_points=2000, math=library.MATH_FAST)
