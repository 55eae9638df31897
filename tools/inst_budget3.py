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
VARIANTS = [("one escaping frequency (staging, one list pass)", [30.0], 200, None),
            ("174 escaping frequencies (staging + list)", np.full(174, 30.0), 200, None),
            ("config 3 at n_points = 2 (one wave-iteration per item)", f174, 2, None),
            ("n_points = 18 (two)", f174, 18, None),
            ("n_points = 34 (three)", f174, 34, None),
            ("n_points = 66 (five)", f174, 66, None),
            ("n_points = 130 (nine)", f174, 130, None),
            ("config 3: n_points = 200 (thirteen)", f174, 200, None),
            ("config 3, no point queued (well_conditioned = 0)", f174, 200, 0.0),
            ("n_points = 392 (twenty-five)", f174, 392, None)]
for name, freq, n, wc in VARIANTS:
    f = torch.as_tensor(np.asarray(freq, dtype=np.float64), device=dev)
    library.set_option("well_conditioned", 1e-5 if wc is None else wc)
    ms = []
    for _ in range(REPS):
        out = library.vertical_forward_operator(f, *t, "O", n)
        ms.append(ctx.last_kernel_ms())
    torch.cuda.synchronize()
    print(json.dumps({"variant": name, "n_freq": int(f.numel()), "n_points": n, "well_conditioned": wc, "reps": REPS,
                      "kernel_ms": min(ms[1:]), "finite": float(torch.isfinite(out).double().mean()), "options": OPTIONS}), flush=True)
library.set_option("well_conditioned", 1e-5)
