#!/usr/bin/env python3
"""Instruction budget of the short-grid O kernel on config 3 (10 000 x 174, O/200): one launch per variant, each
repeated REPS times, so that a profiler pass (tools/inst_budget3.sh: kernel trace, then SQ_INSTS_* in their own passes)
sees the dispatches of `vfo_short_kernel<256>` in this order.  The differences between the variants split the kernel's
instructions and time into staging, candidate list, per-item set-up, loop, queue.  Prints the variant list."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
import _options
OPTIONS = _options.apply()
REPS = 3
dev = torch.device("cuda", 0); ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t = [torch.as_tensor(x, device=dev) for x in (den, bmag, bpsi, alt)]
f174 = synth.sounder_frequencies(3)
def lanes_for(n):                      # the library's rule (Knobs::short_lanes = 0): eight lanes per pair up to 256 points
    forced = OPTIONS.get("short_lanes", 0)
    return int(forced) if forced else (8 if n <= 256 else 16)


def iterations(n):
    return -(-n // lanes_for(n))


VARIANTS = [("one escaping frequency (staging, one list pass)", [30.0], 200, None),
            ("174 escaping frequencies (staging + list)", np.full(174, 30.0), 200, None)]
for n in (2, 18, 34, 66, 130, 200, 256):
    VARIANTS.append((f"{'config 3: ' if n == 200 else ''}n_points = {n} ({iterations(n)} wave-iterations per item, {lanes_for(n)} lanes per pair)", f174, n, None))
VARIANTS.append(("config 3, no point queued (well_conditioned = 0)", f174, 200, 0.0))
for name, freq, n, wc in VARIANTS:
    f = torch.as_tensor(np.asarray(freq, dtype=np.float64), device=dev)
    library.set_option("well_conditioned", 1e-5 if wc is None else wc)
    ms = []
    for _ in range(REPS):
        out = library.vertical_forward_operator(f, *t, "O", n)
        ms.append(ctx.last_kernel_ms())
    torch.cuda.synchronize()
    print(json.dumps({"variant": name, "n_freq": int(f.numel()), "n_points": n, "well_conditioned": wc, "reps": REPS,
                      "iterations": iterations(n), "lanes": lanes_for(n),
                      "kernel_ms": min(ms[1:]), "finite": float(torch.isfinite(out).double().mean()), "options": OPTIONS}), flush=True)
library.set_option("well_conditioned", 1e-5)
