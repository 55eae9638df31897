#!/usr/bin/env python3
"""Short-grid O kernel: workgroup size x workgroups per CU x staged levels, same box (one child per setting).
    python tools/ab_short_geom.py "lib.so[;option=value...]" ...   (lib '-' = the in-tree build; options of prhf_ctx_set_option)
Prints the kernel time of BASELINE config 3 (and O/500, O/1000, the config-5 O/200 slice shape) and whether the
outputs equal the first setting's bit for bit."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def child(tag, opts):
    import torch
    from pyrayhf_amd import library, synth, _native
    dev = torch.device("cuda", 0)
    ctx = _native.context(0)
    for k, v in opts.items():
        ctx.set_option(k, float(v))
    res = {"tag": tag}
    alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
    for name, freq, rows, n in (("c3_o200", synth.sounder_frequencies(1), 10000, 200), ("o500", synth.sounder_frequencies(1), 10000, 500),
                                ("o1000", synth.sounder_frequencies(1), 10000, 1000), ("o200_f512", synth.sounder_frequencies(5), 2500, 200)):
        t = [torch.as_tensor(x, device=dev) for x in (freq, den[:rows], bmag[:rows], bpsi[:rows], alt)]
        ms = []
        for r in range(14):
            out = library.vertical_forward_operator(*t, "O", n, sync=True)
            if r >= 2:
                ms.append(ctx.last_kernel_ms())
        res[name] = round(float(np.median(ms)), 4)
        np.save(os.path.join(ROOT, "gpurun_out", f"geom_{tag}_{name}.npy"), out.cpu().numpy())
    print(json.dumps(res), flush=True)

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    child(sys.argv[2], json.loads(sys.argv[3]))
    sys.exit(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
rows = []
for i, spec in enumerate(sys.argv[1:]):
    lib, *opts = spec.split(";")
    env = dict(os.environ)
    if lib != "-":
        env["PRHF_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f"s{i}",
                          json.dumps({o.split("=")[0]: float(o.split("=")[1]) for o in opts})],
                         env=env, check=True, capture_output=True, text=True).stdout
    r = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    r["spec"] = spec
    for name in ("c3_o200", "o500", "o1000", "o200_f512"):
        a = np.load(os.path.join(ROOT, "gpurun_out", f"geom_s0_{name}.npy"))
        b = np.load(os.path.join(ROOT, "gpurun_out", f"geom_s{i}_{name}.npy"))
        r[name + "_same"] = bool(np.array_equal(a, b, equal_nan=True))
    rows.append(r)
    print(json.dumps(r), flush=True)
for name in ("c3_o200", "o500", "o1000", "o200_f512"):
    for f in os.listdir(os.path.join(ROOT, "gpurun_out")):
        if f.startswith("geom_") and f.endswith(name + ".npy"):
            os.remove(os.path.join(ROOT, "gpurun_out", f))
