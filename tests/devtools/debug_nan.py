import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyrayhf_amd import library as lib
from oracle import vfo_numpy as orc
g = np.load('tests/golden/g5_chapman64.npz')
np.seterr(all='ignore')
vx = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 2000, math=lib.MATH_FAST)
want = g["X_2000_vh"]
bad = np.argwhere(np.isnan(vx) != np.isnan(want))
print("mask diffs", bad[:10].tolist(), len(bad))
for p, f in bad[:3]:
    print("pair", p, f, "freq", g["freq"][f], "got", vx[p, f], "want", want[p, f])
    cap = orc.stage_capture(g["freq"][f:f+1], g["den"][p], g["bmag"][p], g["bpsi"][p], g["alt"], "X", 2000)
    X, Y, psi = cap["X"][0], cap["Y"][0], cap["bpsi"][0]
    mu_f, mup_f = lib.find_mu_mup(X, Y, psi, "X", math=lib.MATH_FAST)
    mu_r, mup_r = cap["mu"][0], cap["mup"][0]
    odd = np.argwhere(~np.isfinite(mup_f) | (np.abs(mup_f - mup_r) > 1e-6 * np.abs(mup_r))).ravel()
    print(" odd points", odd[:10], len(odd))
    for i in odd[:5]:
        print("  i", i, "X", X[i], "Y", Y[i], "psi", psi[i], "fast", mu_f[i], mup_f[i], "ref", mu_r[i], mup_r[i])
    print(" sum fast", np.nansum(mup_f * cap["dist"][0]), "sum ref", np.nansum(mup_r * cap["dist"][0]))
    print(" crit", cap["crit_height"][0, 0], "K", int(np.argmax(g["den"][p])))
