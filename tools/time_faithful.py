import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyrayhf_amd import library, synth, _native
dev = torch.device("cuda", 0); ctx = _native.context(0)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(3), den, bmag, bpsi, alt)]
for n in (200, 2000):
    ms = []
    for r in range(4):
        library.vertical_forward_operator(*t, "O", n, math=0); ms.append(ctx.last_kernel_ms())
    print(os.environ.get("PRHF_LIB", "default"), "O faithful n_points", n, min(ms[1:]))
