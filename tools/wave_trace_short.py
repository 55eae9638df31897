"""Per-workgroup timeline of the short-grid kernel on config 3 (10 000 x 174, O mode, n_points = 200) from a
-DPRHF_TRACE build (PRHF_LIB points at it): eight wall-clock marks per wave and block -
0 block start, 1 nodes staged, 2 list and heights made, 3 wave out of items, 4 behind the barrier,
5 wave's share of the queue done, 6 results stored, 7 behind the block's last barrier."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
path = "/tmp/prhf_trace_short.bin"
os.environ["PRHF_TRACE_FILE"] = path
from pyrayhf_amd import library, synth
dev = torch.device("cuda", 0)
n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 200
compact = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # 1: four 4-wave workgroups per CU (the default geometry)
W, slots = (4, 1024) if compact else (8, 512)
library.set_option("short_compact", compact)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(3), den, bmag, bpsi, alt)]
for _ in range(2):
    library.vertical_forward_operator(*t, "O", n_points)
w = np.fromfile(path, dtype=np.uint64)[: 10000 * W * 8].reshape(-1, W, 8).astype(np.float64) / 100.0   # us; (block, wave, mark)
w -= w[:, :, 0].min()
d = {}
names = ["nodes", "max+list", "items (this wave)", "wait at barrier A", "queue (this wave)", "wait B + sums", "last barrier"]
for k, name in enumerate(names):
    seg = w[:, :, k + 1] - w[:, :, k]
    d[name] = {"mean_us": round(float(seg.mean()), 2), "p90_us": round(float(np.percentile(seg, 90)), 2)}
life = w[:, :, 7].max(axis=1) - w[:, :, 0].min(axis=1)
print(json.dumps({"blocks": int(w.shape[0]), "n_points": n_points, "waves_per_workgroup": W, "kernel_us": round(float(w[:, :, 7].max()), 1),
                  "block_life_us": {"mean": round(float(life.mean()), 2), "p10": round(float(np.percentile(life, 10)), 2),
                                    "p90": round(float(np.percentile(life, 90)), 2)},
                  "phases_per_wave": d,
                  "sum_of_block_lives_over_kernel_x_slots": round(float(life.sum() / (w[:, :, 7].max() * slots)), 3)}))
