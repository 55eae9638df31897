"""Kernel and per-call time of the small / short-grid configurations (1, 2, 3) under the current knobs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0); ctx = _native.context(0)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g4_day_night.npz"))
day = [g["Day_" + k] for k in ("den", "bmag", "bpsi", "alt")]
f174 = synth.sounder_frequencies(1)
def run(name, freq, den, bmag, bpsi, alt, mode, n, device_inputs, reps=30):
    args = [freq, den, bmag, bpsi, alt]
    if device_inputs:
        args = [torch.as_tensor(x, device=dev) for x in args]
    ms = []
    for _ in range(5):
        library.vertical_forward_operator(*args, mode, n); ms.append(ctx.last_kernel_ms())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        library.vertical_forward_operator(*args, mode, n)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
    print(json.dumps({"case": name, "kernel_us": 1e3 * min(ms[1:]), "call_us": 1e6 * wall}), flush=True)
run("config 1 (O/200), device inputs", f174, *day, "O", 200, True)
run("config 1 (O/200), host inputs", f174, *day, "O", 200, False)
run("config 2 (X/20000), device inputs", f174, *day, "X", 20000, True)
run("config 2 (X/20000), host inputs", f174, *day, "X", 20000, False)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
run("config 3 (10000 x 174, O/200), device inputs", f174, den, bmag, bpsi, alt, "O", 200, True, reps=5)
