import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pyrayhf_amd import library, synth, _native
g = np.load("tests/golden/g4_day_night.npz")
dev = torch.device("cuda", 0); ctx = _native.context(0)
def med(args, mode, n, math=None):
    ms = []
    for r in range(40):
        library.vertical_forward_operator(*args, mode, n, sync=True, math=math)
        ms.append(ctx.last_kernel_ms())
    return round(float(np.median(ms[10:])) * 1e3, 1)
f174 = synth.sounder_frequencies(1)
day = [torch.as_tensor(x, device=dev) for x in (f174, g["Day_den"], g["Day_bmag"], g["Day_bpsi"], g["Day_alt"])]
tiny = [torch.as_tensor(np.asarray(x, dtype=float), device=dev) for x in ([1.0, 2.0, 10.0], [0, 0.5e12, 1e12], [5e-5]*3, [60.0]*3, [100, 200, 300])]
one = [torch.as_tensor(x, device=dev) for x in (f174[:1], g["Day_den"], g["Day_bmag"], g["Day_bpsi"], g["Day_alt"])]
print("tiny 3-level profile x 3 freqs, n=2   :", med(tiny, "O", 2), "us")
print("Day x 1 freq,  O n=2                  :", med(one, "O", 2), "us")
print("Day x 174,     O n=2                  :", med(day, "O", 2), "us")
print("Day x 174,     O n=200 faithful       :", med(day, "O", 200), "us")
print("Day x 174,     O n=200 fast           :", med(day, "O", 200, library.MATH_FAST), "us")
print("Day x 174,     X n=200 fast           :", med(day, "X", 200), "us")
print("Day x 174,     X n=2000               :", med(day, "X", 2000), "us")
print("Day x 174,     X n=20000              :", med(day, "X", 20000), "us")
