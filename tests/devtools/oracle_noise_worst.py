"""Is the fast tier's worst X-mode deviation (2.9e-10 on five config-4 profiles) above the reference algorithm's own
response to +-1 ulp input jitter?  (C oracle, CPU only.)"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import vfo_c
from pyrayhf_amd import synth
rows = np.array([2731, 2120, 2098, 709, 107]); cols = [14, 4, 80, 16, 2]
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=rows)
freq = synth.sounder_frequencies(4)
base = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", 20000)
rng = np.random.default_rng(0)
worst = np.zeros_like(base)
for _ in range(16):
    j = lambda a: np.nextafter(a, a + rng.choice([-1.0, 1.0], size=a.shape) * np.abs(a) - 0 * a + np.where(a == 0, 1, 0))
    v = vfo_c.virtual_heights_batch(j(freq), j(den), j(bmag), j(bpsi), alt, "X", 20000)
    worst = np.fmax(worst, np.abs(v - base) / np.abs(base))
for r, c in zip(range(5), cols):
    print(f"profile {rows[r]} f[{c}] = {freq[c]:.4f} MHz: jitter response {worst[r, c]:.2e}; row max {np.nanmax(worst[r]):.2e}")
