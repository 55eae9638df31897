"""Timeline of a single-profile launch (config 2) from a -DPRHF_TRACE build: when do the workgroups start,
how long is staging, when does the last one finish."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
path = "/tmp/prhf_trace.bin"
os.environ["PRHF_TRACE_FILE"] = path
from pyrayhf_amd import library, synth, _native
g = np.load(os.path.join(ROOT, "tests", "golden", "g4_day_night.npz"))
day = [g["Day_" + k] for k in ("den", "bmag", "bpsi", "alt")]
f174 = synth.sounder_frequencies(1)
dev = torch.device("cuda", 0)
t = [torch.as_tensor(x, device=dev) for x in (f174, *day)]
for mode, n in (("X", 20000), ("O", 200), ("O", 20000)):
    for _ in range(3):
        library.vertical_forward_operator(*t, mode, n)
    w = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 6).astype(np.float64) / 100.0
    t0 = w[:, :, 0].min()
    print(json.dumps({"case": f"{mode}/{n}", "workgroups": int(w.shape[0]), "kernel_ms_events": _native.context(0).last_kernel_ms(),
                      "first_start_us": 0.0, "last_start_us": float(w[:, :, 0].max() - t0),
                      "staging_us_mean": float((w[:, :, 2] - w[:, :, 0]).mean()), "staging_us_max": float((w[:, :, 2] - w[:, :, 0]).max()),
                      "argmax_known_us": float((w[:, 0, 3] - w[:, 0, 0]).mean()), "nodes_staged_us": float((w[:, 0, 4] - w[:, 0, 0]).mean()),
                      "running_max_done_us": float((w[:, 0, 5] - w[:, 0, 0]).mean()),
                      "work_after_staging_us_mean": float((w[:, :, 1] - w[:, :, 2]).mean()),
                      "work_after_staging_us_by_wave_max": [round(float(x), 1) for x in (w[:, :, 1] - w[:, :, 2]).max(axis=0)],
                      "last_end_us": float(w[:, :, 1].max() - t0)}))
