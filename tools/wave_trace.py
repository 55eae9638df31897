"""Per-wave timeline of the fused kernel (needs a -DPRHF_TRACE build of libprhf.so, PRHF_LIB pointing at it).
Prints how full the wave slots of a workgroup are over its life and how the launch drains."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
path = "/tmp/prhf_trace.bin"
os.environ["PRHF_TRACE_FILE"] = path
from pyrayhf_amd import library, synth
dev = torch.device("cuda", 0)
alt, den, bmag, bpsi = synth.chapman_profiles(12500, 20260004)
t = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(4), den, bmag, bpsi, alt)]
for _ in range(2):
    library.vertical_forward_operator(*t, "X", 20000)
w = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 6).astype(np.float64) / 100.0  # us (100 MHz wall clock): start, end, staged
t0 = w[:, :, 0].min()
start, end = w[:, :, 0] - t0, w[:, :, 1] - t0
wg_start, wg_end = start.min(axis=1), end.max(axis=1)
life = wg_end - wg_start
fill = (end - start).sum(axis=1) / (8 * life)
print(json.dumps({"workgroups": int(w.shape[0]), "kernel_us": float(wg_end.max()),
                  "wg_life_us": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "p90": float(np.percentile(life, 90)), "max": float(life.max())},
                  "wave_slot_fill_within_wg": {"mean": float(fill.mean()), "p10": float(np.percentile(fill, 10)), "weighted": float(((end - start).sum()) / (8 * life.sum()))},
                  "first_wave_done_frac_of_life": float(((end.min(axis=1) - wg_start) / life).mean()),
                  "median_wave_done_frac_of_life": float(((np.median(end, axis=1) - wg_start) / life).mean())}))
print("mean finish time of wave w as a fraction of its workgroup's life:",
      [round(float(x), 3) for x in ((end - wg_start[:, None]) / life[:, None]).mean(axis=0)])
print("staging time per workgroup [us]: mean %.2f  p90 %.2f" % ((w[:, :, 2] - w[:, :, 0]).max(axis=1).mean(),
                                                                  np.percentile((w[:, :, 2] - w[:, :, 0]).max(axis=1), 90)))
print("mean start delay of wave w [us]:", [round(float(x), 2) for x in (start - wg_start[:, None]).mean(axis=0)])
# resident workgroups over time
edges = np.linspace(0, wg_end.max(), 41)
res = [(np.minimum(wg_end, b) - np.maximum(wg_start, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident workgroups per 1/40 of the launch:", [round(x) for x in res])
waves = [(np.minimum(end, b) - np.maximum(start, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident waves per 1/40 of the launch:", [round(x) for x in waves])
