"""Where does the fast tier's worst X-mode deviation come from?  The five worst config-4 profiles under each
arithmetic variant of the library at PRHF_LIB, against the C oracle."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import vfo_c
from pyrayhf_amd import library, synth
rows = np.array([2731, 2120, 2098, 709, 107])
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=rows)
freq = synth.sounder_frequencies(4)
want = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", 20000)
ok = np.isfinite(want)
for name, math in (("default (fast)", None), ("faithful", library.MATH_FAITHFUL)):
    got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", 20000, math=math)
    err = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
    print(os.environ.get("PRHF_LIB", "default lib"), name, "max", err.max(), "median", np.median(err), flush=True)
