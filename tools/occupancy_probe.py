import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0); ctx = _native.context(0)
freq = synth.sounder_frequencies(4)
alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, 4096))
# make every profile identical so that block times are equal
den[:] = den[0]; bmag[:] = bmag[0]; bpsi[:] = bpsi[0]
T = {k: torch.as_tensor(v, device=dev) for k, v in (("freq", freq), ("alt", alt), ("den", den), ("bmag", bmag), ("bpsi", bpsi))}
for P in (128, 256, 512, 768, 1024, 2048, 4096):
    ms = []
    for r in range(6):
        library.vertical_forward_operator(T["freq"], T["den"][:P], T["bmag"][:P], T["bpsi"][:P], T["alt"], "X", 20000, sync=True)
        ms.append(ctx.last_kernel_ms())
    print(P, "profiles (blocks):", round(float(np.median(ms[2:])), 3), "ms")
