#!/usr/bin/env python3
"""Short-grid O kernel on config 3 (10 000 x 174, O/200) and on the config-5 O/200 slice under option settings, interleaved
on one box: python tools/ab_short_opts.py "short_prio=1" "short_prio=5" "short_prio=4,short_order=0" ...
(each argument one setting, name=value pairs separated by commas; the first is the reference).  Median kernel time per
setting and round (HIP events), results compared bit for bit with the first setting."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrayhf_amd import library, synth, _native, dist as pdist
from bench import CONFIG5_SEGMENTS
DEFAULTS = {"short_prio": 4, "short_order": 1, "short_lanes": 0, "short_compact": 1}
settings = [dict(item.split("=") for item in arg.split(",")) for arg in sys.argv[1:]] or [{}]
dev = torch.device("cuda", 0)
ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t3 = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(3), den, bmag, bpsi, alt)]
rows, segs = pdist.shard_segments(CONFIG5_SEGMENTS, 8, 0)
a5, d5, b5, p5 = synth.chapman_profiles(50000, 20260005, rows=rows)
t5 = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(5), d5, b5, p5, a5)]
p0, p1 = segs[0][0], segs[0][1]
cases = [("config 3", lambda: library.vertical_forward_operator(*t3, "O", 200)),
         ("config-5 O/200 slice", lambda: library.vertical_forward_operator(t5[0], t5[1][p0:p1], t5[2][p0:p1], t5[3][p0:p1], t5[4], "O", 200)),
         ("config-5 shard", lambda: library.vertical_forward_operator_mixed(*t5, segs))]
for name, fn in cases:
    times = {i: [] for i in range(len(settings))}
    outs = {}
    for rnd in range(3):
        for i, st in enumerate(settings):
            for k, v in {**DEFAULTS, **{k: float(v) for k, v in st.items()}}.items():
                library.set_option(k, v)
            ms = []
            for _ in range(9):
                out = fn()
                ms.append(ctx.last_kernel_ms())
            times[i].append(float(np.median(ms[2:])))
            outs[i] = out
    for i, st in enumerate(settings):
        same = bool(torch.equal(torch.nan_to_num(outs[i], nan=-1.0), torch.nan_to_num(outs[0], nan=-1.0)))
        print(json.dumps({"case": name, "setting": st, "kernel_ms": times[i], "best": min(times[i]),
                          "vs_first": min(times[i]) / min(times[0]), "same_bits_as_first": same}), flush=True)
for k, v in DEFAULTS.items():
    library.set_option(k, v)
