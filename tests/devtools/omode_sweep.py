"""O mode against the NumPy oracle (bit-identical to the reference on every fixture) on the first profiles of
BASELINE config 3 - no noise floor exists for them beyond fixture G10's 64 rows: report the distribution.
Also the plain-C restatement, to show where IT stands against the same oracle."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import vfo_c, vfo_numpy
from pyrayhf_amd import library, synth
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003, rows=slice(0, 400))
freq = synth.sounder_frequencies(3)
for n, rows in ((200, 400), (2000, 200), (20000, 40)):
    want = vfo_numpy.virtual_heights_batch(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n)
    runs = [("default", lambda: library.vertical_forward_operator(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n)),
            ("reference order everywhere", lambda: library.vertical_forward_operator(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n, math=library.MATH_FAITHFUL)),
            ("reduced algebra everywhere", lambda: library.vertical_forward_operator(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n, math=library.MATH_FAST)),
            ("C restatement (oracle/vfo_oracle.c)", lambda: vfo_c.virtual_heights_batch(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n))]
    for name, fn in runs:
        got = fn()
        mask = int((np.isnan(got) != np.isnan(want)).sum())
        ok = np.isfinite(want) & np.isfinite(got)
        err = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        print(json.dumps({"n_points": n, "profiles": rows, "arithmetic": name, "against": "oracle/vfo_numpy.py", "pairs": int(ok.sum()), "mask_diffs": mask,
                          "median": float(np.median(err)), "p99": float(np.percentile(err, 99)), "max": float(err.max()),
                          "within_1e-6": float((err <= 1e-6).mean()), "bit_identical": float((err == 0).mean())}), flush=True)
