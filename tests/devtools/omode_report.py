"""O-mode parity against the reference-generated fixtures with noise floors (G4, G5, G10): per fixture and
arithmetic setting, the share of finite pairs within 1e-6, the worst pair, and how many pairs exceed
max(1e-6, 4 * noise) with the noise taken per pair (window 0) or maximised over +-2 frequencies (window 2)."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np

from conftest import load_golden
from parity import effective_noise, rel_err
from pyrayhf_amd import library

cases = []
g4 = load_golden("g4_day_night.npz")
for which in ("Day", "Night"):
    for n in (200, 2000, 20000):
        cases.append((f"G4 {which} O/{n}", g4["freq"], g4[f"{which}_den"], g4[f"{which}_bmag"], g4[f"{which}_bpsi"],
                      g4[f"{which}_alt"], n, g4[f"{which}_O_{n}_vh"], g4[f"{which}_O_{n}_noise"]))
g5 = load_golden("g5_chapman64.npz")
cases.append(("G5 O/200", g5["freq"], g5["den"], g5["bmag"], g5["bpsi"], g5["alt"], 200, g5["O_200_vh"], g5["O_200_noise"]))
g10 = load_golden("g10_config3_rows.npz")
cases.append(("G10 O/200", g10["freq"], g10["den"], g10["bmag"], g10["bpsi"], g10["alt"], 200, g10["O_200_vh"],
              g10["O_200_noise"]))
for name, freq, den, bmag, bpsi, alt, n, want, noise in cases:
    for label, math in (("default", None), ("reference order", library.MATH_FAITHFUL), ("reduced", library.MATH_FAST)):
        got = library.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", n, math=math)
        err, ok = rel_err(got, want)
        rec = {"fixture": name, "arithmetic": label, "pairs": int(ok.sum()),
               "mask_diffs": int((np.isnan(got) != np.isnan(want)).sum()),
               "bit_identical": float((got[ok] == want[ok]).mean()),
               "within_1e-6": float((err[ok] <= 1e-6).mean()), "median": float(np.median(err[ok])),
               "max": float(err[ok].max())}
        for w in (0, 2):
            lim = np.maximum(1e-6, 4.0 * effective_noise(noise, w))
            rec[f"over_4noise_w{w}"] = int((ok & (err > lim)).sum())
        lim1 = np.maximum(1e-6, 1.0 * effective_noise(noise, 0))
        rec["over_1noise_w0"] = int((ok & (err > lim1)).sum())
        print(json.dumps(rec), flush=True)
