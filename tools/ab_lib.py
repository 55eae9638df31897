#!/usr/bin/env python3
"""Kernel times of a few workload shapes for two builds of the library on the same box:
    python tools/ab_lib.py build/ab/libprhf_X.so [build/ab/libprhf_Y.so ...]
(the in-tree pyrayhf_amd/libprhf.so is always the first column).  One child process per build."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def child():
    import torch
    from pyrayhf_amd import library, synth, _native, dist as pdist
    from bench import CONFIG5_SEGMENTS
    dev = torch.device("cuda", 0)
    ctx = _native.context(0)
    res = {}
    def run(name, freq, alt, den, bmag, bpsi, mode, n, reps=6):
        t = [torch.as_tensor(x, device=dev) for x in (freq, den, bmag, bpsi, alt)]
        ms = []
        for r in range(reps + 2):
            library.vertical_forward_operator(*t, mode, n, sync=True)
            if r >= 2:
                ms.append(ctx.last_kernel_ms())
        res[name] = round(float(np.median(ms)), 4)
    f174, f256, f512 = synth.sounder_frequencies(1), synth.sounder_frequencies(4), synth.sounder_frequencies(5)
    alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
    run("config3 O/200 10000x174", f174, alt, den, bmag, bpsi, "O", 200, reps=10)
    run("O/500 10000x174", f174, alt, den, bmag, bpsi, "O", 500)
    run("X/200 10000x174", f174, alt, den, bmag, bpsi, "X", 200, reps=10)
    run("X/1000 10000x174", f174, alt, den, bmag, bpsi, "X", 1000)
    run("O/2000 5000x512", f512, alt, den[:5000], bmag[:5000], bpsi[:5000], "O", 2000)
    run("X/2000 7500x512", f512, alt, den[:7500], bmag[:7500], bpsi[:7500], "X", 2000)
    run("O/20000 1000x174", f174, alt, den[:1000], bmag[:1000], bpsi[:1000], "O", 20000, reps=3)
    alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, 12500))
    run("config4 shard X/20000 12500x256", f256, alt, den, bmag, bpsi, "X", 20000, reps=4)
    rows, segs = pdist.shard_segments(CONFIG5_SEGMENTS, 8, 0)
    alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=rows)
    tt = [torch.as_tensor(x, device=dev) for x in (f512, den, bmag, bpsi, alt)]
    ms = []
    for r in range(8):
        library.vertical_forward_operator_mixed(*tt, segs)
        torch.cuda.synchronize()
        if r >= 2:
            ms.append(ctx.last_kernel_ms())
    res["config5 shard"] = round(float(np.median(ms)), 4)
    print(json.dumps(res), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child()
    sys.exit(0)
libs = [os.path.join(ROOT, "pyrayhf_amd", "libprhf.so")] + [os.path.abspath(p) for p in sys.argv[1:]]
rows = []
for lib in libs:
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, PRHF_LIB=lib),
                         check=True, capture_output=True, text=True).stdout
    rows.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
for k in rows[0]:
    print(f"{k:36s}" + "".join(f"{r[k]:10.4f}" for r in rows) + ("" if len(rows) < 2 else f"   x{rows[1][k] / rows[0][k]:.3f}"))
