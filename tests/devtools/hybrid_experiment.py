"""CPU experiment: O mode evaluated with the reduced (fast-tier) algebra wherever 1 - X > thr and with the
reference's operation order elsewhere.  How far is that from the reference, in units of its own noise floor?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vfo_numpy as orc
from parity import rel_err, effective_noise


def mup_fast(X, Y, psi_deg, mode):
    sgn = 1.0 if mode == "O" else -1.0
    S2 = np.sin(np.deg2rad(psi_deg)) ** 2
    Y2 = Y * Y
    Xm1 = 1.0 - X
    YT2 = Y2 * S2
    YL2 = Y2 - YT2
    h = 0.5 * YT2
    h2 = h * h
    t = YL2 * Xm1
    alpha = h2 + t * Xm1
    rbeta = 1.0 / np.sqrt(alpha)
    beta = alpha * rbeta
    D = (Xm1 - h) + sgn * beta
    XXm1 = X * Xm1
    N = D - XXm1
    w = 1.0 / np.sqrt(N * D)
    Nw = N * w
    mu = np.abs(Nw)
    rD = Nw * w
    q = XXm1 * rD
    two_X = X + X
    inner = (0.5 * sgn) * ((h2 - two_X * t) * rbeta + beta) - (X + h)
    half = q * inner + (two_X * X - X)
    mup = mu - np.copysign(w, D) * half
    mup[~(q > -3.3306690738754696e-16)] = np.nan
    return mup


def evaluate(freq, den, bmag, bpsi, alt, mode, n_points, thr):
    with np.errstate(all="ignore"):
        f_hz = freq * 1e6
        cols = orc.stretched_columns(f_hz, den, bmag, bpsi, alt, mode, n_points)
        X = orc.ratio_X(cols["den"], cols["freq"])
        Y = orc.ratio_Y(cols["freq"], cols["bmag"])
        _, mf = orc.phase_group_index(X, Y, cols["bpsi"], mode)
        mq = mup_fast(X, Y, cols["bpsi"], mode)
        out = {}
        for name, m in (("faithful", mf), ("fast", mq), ("hybrid", np.where(1.0 - X > thr, mq, mf))):
            tot = np.nansum(m * cols["dist"], axis=1)
            tot[tot == 0] = np.nan
            out[name] = tot + np.min(alt)
        out["share_faithful"] = float(np.mean((1.0 - X <= thr)[np.isfinite(X)]))
        return out


def report(tag, res, want, noise):
    for name in ("faithful", "fast", "hybrid"):
        err, ok = rel_err(res[name], want)
        ne = effective_noise(noise)
        ratio = np.where(ok, err / np.maximum(1e-6 / 4.0, ne), 0.0)          # in units of the 4x rule's denominator
        print(f"{tag:28s} {name:8s} max err {err.max():.2e}  worst err/noise_eff {ratio.max():6.2f}  "
              f"within 1e-6: {np.mean(err[ok] <= 1e-6):.4f}")


if __name__ == "__main__":
    thr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
    g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_day_night.npz"))
    for who in ("Day", "Night"):
        for n in (200, 2000, 20000):
            res = evaluate(g4["freq"], *(g4[f"{who}_{k}"] for k in ("den", "bmag", "bpsi", "alt")), "O", n, thr)
            report(f"G4 {who} O/{n} (faithful share {res['share_faithful']:.3f})", res, g4[f"{who}_O_{n}_vh"], g4[f"{who}_O_{n}_noise"])
    g5 = np.load(os.path.join(ROOT, "tests", "golden", "g5_chapman64.npz"))
    rows = {k: [] for k in ("faithful", "fast", "hybrid")}
    for p in range(g5["den"].shape[0]):
        res = evaluate(g5["freq"], g5["den"][p], g5["bmag"][p], g5["bpsi"][p], g5["alt"], "O", 200, thr)
        for k in rows:
            rows[k].append(res[k])
    report("G5 64 Chapman O/200", {k: np.array(v) for k, v in rows.items()}, g5["O_200_vh"], g5["O_200_noise"])
