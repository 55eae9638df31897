"""O mode, default arithmetic, against the C oracle on config-3 style profiles (there is no noise floor for
synthetic profiles: report the distribution)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import vfo_c
from pyrayhf_amd import library, synth
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003, rows=slice(0, 400))
freq = synth.sounder_frequencies(3)
for n, rows in ((200, 400), (2000, 200), (20000, 60)):
    want = vfo_c.virtual_heights_batch(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n)
    for name, math in (("default", None), ("reference order everywhere", library.MATH_FAITHFUL), ("reduced algebra everywhere", library.MATH_FAST)):
        got = library.vertical_forward_operator(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n, math=math)
        mask = int((np.isnan(got) != np.isnan(want)).sum())
        ok = np.isfinite(want) & np.isfinite(got)
        err = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        print(json.dumps({"n_points": n, "profiles": rows, "arithmetic": name, "pairs": int(ok.sum()), "mask_diffs": mask,
                          "median": float(np.median(err)), "p99": float(np.percentile(err, 99)), "max": float(err.max()),
                          "within_1e-6": float((err <= 1e-6).mean())}), flush=True)
