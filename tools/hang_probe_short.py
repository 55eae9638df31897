"""Short-grid launches with progress prints (run under `timeout` first after touching the short-grid kernels' control
flow): compact and full-size geometry, profiles whose peak lies above the compact arrays (second launch), a profile
of another input shape (general follow-up), tiny queue; every result compared bit for bit with short_compact = 0."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pyrayhf_amd import library, synth
alt, den, bmag, bpsi = synth.chapman_profiles(300, 11)
freq = synth.sounder_frequencies(1)
# rows 5..9: a layer peaking at 560 km (level 480: above the compact arrays); row 20: negative density (an error in
# the reference - not used); row 21: a field angle that turns fast (general kernel's case)
z = (alt[None, :] - 560.0) / 60.0
den[5:10] = 9e11 * np.exp(0.5 * (1.0 - z - np.exp(-z)))
bpsi[21] = 20.0 + 0.5 * (alt - 80.0)
print("peaks", np.argmax(den, axis=1)[[0, 5, 21]], flush=True)
res = {}
for compact in (0, 1):
    library.set_option("short_compact", compact)
    for q in (0, 40):
        library.set_option("short_queue", q)
        for n in (200, 64, 1000):
            print("launch compact", compact, "queue", q, "n", n, flush=True)
            vh = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", n)
            print("done", int(np.isfinite(vh).sum()), flush=True)
            key = (q, n)
            if compact == 0:
                res[key] = vh
            else:
                same = np.array_equal(vh, res[key], equal_nan=True)
                print("same as full-size geometry:", same, flush=True)
                if not same:
                    bad = np.argwhere((vh != res[key]) & ~(np.isnan(vh) & np.isnan(res[key])))
                    print("differs at", bad[:10].tolist(), flush=True)
                    sys.exit(1)
print("OK", flush=True)
