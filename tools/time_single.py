"""Latency of the single-profile configurations (BASELINE configs 1 and 2) from device-resident inputs."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
import _options
opts = _options.apply()                 # PRHF_TOOL_OPTIONS="name=value,..."
dev = torch.device("cuda", 0); ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(4, 7)
f = synth.sounder_frequencies(1)
t = [torch.as_tensor(x, device=dev) for x in (f, den[:1], bmag[:1], bpsi[:1], alt)]
out = torch.empty((1, f.size), dtype=torch.float64, device=dev)
for mode, n in (("O", 200), ("X", 20000), ("O", 20000)):
    for _ in range(5):
        library.vertical_forward_operator(*t, mode, n, sync=False, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        library.vertical_forward_operator(*t, mode, n, sync=False, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(json.dumps({"mode": mode, "n_points": n, "us_per_enqueued_call": 1e6 * dt, "kernel_us": 1e3 * ctx.last_kernel_ms(), "options": opts}))
