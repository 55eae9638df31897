#!/usr/bin/env python3
"""Short-grid O kernel: 16 against 8 lanes per pair (context option short_lanes) on one box, interleaved: BASELINE config 3
(10 000 x 174, O/200) and other grid sizes, the O/200 slice of the config-5 shard (2 500 x 512), and the whole shard.
Prints kernel times (median of the repetitions, HIP events) and how far the two results are apart."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrayhf_amd import library, synth, _native, dist as pdist
from bench import CONFIG5_SEGMENTS
dev = torch.device("cuda", 0)
ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t3 = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(3), den, bmag, bpsi, alt)]
rows, segs = pdist.shard_segments(CONFIG5_SEGMENTS, 8, 0)
a5, d5, b5, p5 = synth.chapman_profiles(50000, 20260005, rows=rows)
t5 = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(5), d5, b5, p5, a5)]
p0, p1 = segs[0][0], segs[0][1]


def timed(fn, reps=9):
    ms = []
    for _ in range(reps):
        out = fn()
        ms.append(ctx.last_kernel_ms())
    return float(np.median(ms[2:])), out


cases = [(f"config 3 shape, n_points = {n}", (lambda n=n: library.vertical_forward_operator(*t3, "O", n))) for n in (50, 136, 200, 256, 400, 1000)]
cases.append(("config-5 O/200 slice, 2500 x 512", lambda: library.vertical_forward_operator(t5[0], t5[1][p0:p1], t5[2][p0:p1], t5[3][p0:p1], t5[4], "O", 200)))
cases.append(("config-5 shard, mixed list", lambda: library.vertical_forward_operator_mixed(*t5, segs)))
for name, fn in cases:
    res = {}
    for rnd in range(2):                       # interleaved: 16, 8, 16, 8
        for lanes in (16, 8):
            library.set_option("short_lanes", lanes)
            ms, out = timed(fn)
            res.setdefault(lanes, []).append(ms)
            res[f"out{lanes}"] = out
    a, b = res["out16"], res["out8"]
    same_nan = bool(torch.equal(torch.isnan(a), torch.isnan(b)))
    ok = torch.isfinite(a)
    dev_max = float(((a[ok] - b[ok]).abs() / a[ok].abs()).max()) if ok.any() else 0.0
    print(json.dumps({"case": name, "ms_16_lanes": res[16], "ms_8_lanes": res[8], "ratio_8_over_16": min(res[8]) / min(res[16]),
                      "same_nan_mask": same_nan, "max_relative_difference": dev_max}), flush=True)
library.set_option("short_lanes", 0)
