import json, os, sys, subprocess
code = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pyrayhf_amd import library, synth, _native
g = np.load("tests/golden/g4_day_night.npz")
dev = torch.device("cuda", 0); ctx = _native.context(0)
library.set_option("target_waves", float(os.environ["SWEEP_TARGET_WAVES"]))
t = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(1), g["Day_den"], g["Day_bmag"], g["Day_bpsi"], g["Day_alt"])]
res = {}
for mode, n in (("O", 200), ("X", 2000), ("X", 20000)):
    ms = []
    for r in range(30):
        library.vertical_forward_operator(*t, mode, n, sync=True)
        ms.append(ctx.last_kernel_ms())
    res[f"{mode}{n}"] = round(float(np.median(ms[5:])) * 1e3, 1)
print(os.environ.get("SWEEP_TARGET_WAVES"), res)
'''
for tw in ("1024", "2048", "4096", "8192", "16384"):
    env = dict(os.environ, SWEEP_TARGET_WAVES=tw)
    subprocess.run([sys.executable, "-c", code], env=env)
