"""Config-4 shard rows against the C oracle: NaN-mask and value differences, by profile."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import vfo_c
from pyrayhf_amd import library, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, n))
freq = synth.sounder_frequencies(4)
got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", 20000)
want = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", 20000)
bad = np.isnan(got) != np.isnan(want)
print("pairs", got.size, "finite got", np.isfinite(got).sum(), "finite want", np.isfinite(want).sum(), "mask diffs", bad.sum())
for (p, f) in np.argwhere(bad)[:12]:
    print(f"p {p} f {f} freq {freq[f]:.4f} got {got[p,f]!r} want {want[p,f]!r} den0 {den[p,0]:.3e} den1 {den[p,1]:.3e} "
          f"fH0 {2.799249247e10*bmag[p,0]/1e6:.4f} K {int(np.argmax(den[p]))}")
ok = np.isfinite(got) & np.isfinite(want)
err = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
print("max rel err", err.max(), "median", np.median(err), "count > 1e-9", (err > 1e-9).sum())
w = np.argwhere(ok)[np.argsort(err)[-5:]]
for (p, f) in w:
    print(f"worst p {p} f {f} freq {freq[f]:.4f} got {got[p,f]!r} want {want[p,f]!r}")
