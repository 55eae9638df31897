"""A/B two builds of libprhf.so on the config-4 shard: PRHF_LIB_A / PRHF_LIB_B, each run in a child process."""
import sys, os, subprocess, json
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    from pyrayhf_amd import library, synth
    alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, 12500))
    vh = library.vertical_forward_operator(synth.sounder_frequencies(4), den, bmag, bpsi, alt, "X", 20000)
    np.save(sys.argv[2], vh)
    sys.exit(0)
out = {}
for tag in "AB":
    env = dict(os.environ, PRHF_LIB=os.environ["PRHF_LIB_" + tag])
    path = f"/tmp/vh_{tag}.npy"
    subprocess.run([sys.executable, __file__, "child", path], env=env, check=True)
    out[tag] = np.load(path)
a, b = out["A"], out["B"]
bad = np.isnan(a) != np.isnan(b)
print("finite A", np.isfinite(a).sum(), "finite B", np.isfinite(b).sum(), "mask diffs", bad.sum())
rows = np.unique(np.argwhere(bad)[:, 0])
print("rows with diffs", rows.size, rows[:20].tolist())
sys.path.insert(0, os.getcwd())
from oracle import vfo_c
from pyrayhf_amd import synth
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, 12500))
freq = synth.sounder_frequencies(4)
pick = rows[:6]
if pick.size:
    want = vfo_c.virtual_heights_batch(freq, den[pick], bmag[pick], bpsi[pick], alt, "X", 20000)
    for i, p in enumerate(pick):
        fs = np.flatnonzero(bad[p])
        print(f"row {p}: diff freqs {fs[:6].tolist()} ({fs.size}) A {a[p, fs[:3]].tolist()} B {b[p, fs[:3]].tolist()} oracle {want[i, fs[:3]].tolist()} "
              f"den0 {den[p,0]:.3e} den1 {den[p,1]:.3e} fH0 {2.799249247e10*bmag[p,0]/1e6:.4f} K {int(np.argmax(den[p]))}")
ok = np.isfinite(a) & np.isfinite(b)
print("max |A-B|/|A|", (np.abs(a[ok] - b[ok]) / np.abs(a[ok])).max())
