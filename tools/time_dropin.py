"""Latency of the drop-in call as a PyRayHF user makes it: NumPy arrays in, NumPy array out, one profile."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrayhf_amd import library, synth
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _options
opts = _options.apply()                                   # PRHF_TOOL_OPTIONS="local_chunks=0" for A/B runs; "timing=1":
                                                          # kernel_us from the library's events (3.5 us more per call)
alt, den, bmag, bpsi = synth.chapman_profiles(4, 99)
freq = synth.sounder_frequencies(1)
for mode, n in (("O", 200), ("X", 200), ("O", 2000), ("X", 20000)):
    for _ in range(5):
        library.vertical_forward_operator(freq, den[0], bmag[0], bpsi[0], alt, mode, n)
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        vh = library.vertical_forward_operator(freq, den[0], bmag[0], bpsi[0], alt, mode, n)
    dt = (time.perf_counter() - t0) / reps
    # the same call with the foreign function replaced by a no-op: what the Python binding itself costs
    from pyrayhf_amd import _native
    ctx = _native.host_context(None)
    real = ctx.vfo_batch
    ctx.vfo_batch = lambda *a, **k: 0
    t1 = time.perf_counter()
    for _ in range(reps):
        library.vertical_forward_operator(freq, den[0], bmag[0], bpsi[0], alt, mode, n)
    binding = (time.perf_counter() - t1) / reps
    ctx.vfo_batch = real
    try:
        kernel_us = 1e3 * library.last_kernel_ms()
    except Exception:                      # option timing=0: host calls record no events
        kernel_us = None
    print(json.dumps({"call": f"vertical_forward_operator(174 freqs, 1 profile, '{mode}', {n}) on NumPy arrays", "us_per_call": 1e6 * dt, "python_binding_us": 1e6 * binding,
                      "kernel_us": kernel_us, "finite": int(np.isfinite(vh).sum()), "options": opts}), flush=True)
