"""Diagnostics (PRHF_LIB=pyrayhf_amd/libprhf_dbg.so, built with -DPRHF_MARK_FALLBACK): how many pairs of the config-4
shard leave the main loop with a non-finite sum and are re-run by the generic loop."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
from pyrayhf_amd import library, synth
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, 2000))
freq = synth.sounder_frequencies(4)
for mode, n in (("X", 20000), ("X", 2000), ("O", 2000)):
    vh = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n, math=library.MATH_FAST)
    fin = np.isfinite(vh)
    print(json.dumps({"mode": mode, "n_points": n, "pairs": int(vh.size), "reflecting": int(fin.sum()),
                      "fell_back": int((vh[fin] > 1e5).sum())}))
