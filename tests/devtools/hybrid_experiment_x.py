"""CPU experiment, X mode: reduced algebra everywhere vs reference order where mu^2 <= thr, against the reference."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vfo_numpy as orc
from parity import rel_err
from hybrid_experiment import mup_fast          # noqa: E402

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_day_night.npz"))
for who in ("Day", "Night"):
    for n in (2000, 20000):
        den, bmag, bpsi, alt = (g4[f"{who}_{k}"] for k in ("den", "bmag", "bpsi", "alt"))
        with np.errstate(all="ignore"):
            f_hz = g4["freq"] * 1e6
            cols = orc.stretched_columns(f_hz, den, bmag, bpsi, alt, "X", n)
            X = orc.ratio_X(cols["den"], cols["freq"]); Y = orc.ratio_Y(cols["freq"], cols["bmag"])
            mu, mf = orc.phase_group_index(X, Y, cols["bpsi"], "X")
            mq = mup_fast(X, Y, cols["bpsi"], "X")
            for name, m in (("faithful", mf), ("fast", mq), ("hybrid", np.where(mu * mu > thr, mq, mf))):
                tot = np.nansum(m * cols["dist"], axis=1); tot[tot == 0] = np.nan
                err, ok = rel_err(tot + np.min(alt), g4[f"{who}_X_{n}_vh"])
                print(f"{who} X/{n} {name:8s} max err {err.max():.2e} median {np.median(err[ok]):.2e}")
