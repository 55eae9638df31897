"""Per-workgroup timeline of config 3 (10 000 x 174, O mode, n_points = 200) from a -DPRHF_TRACE build
(PRHF_LIB points at it): staging time, block life, how the launch fills and drains."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
path = "/tmp/prhf_trace3.bin"
os.environ["PRHF_TRACE_FILE"] = path
from pyrayhf_amd import library, synth
dev = torch.device("cuda", 0)
alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
t = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(3), den, bmag, bpsi, alt)]
for _ in range(2):
    library.vertical_forward_operator(*t, "O", 200)
w = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 6).astype(np.float64) / 100.0  # us: start, end, staged
t0 = w[:, :, 0].min()
start, end, staged = w[:, :, 0] - t0, w[:, :, 1] - t0, w[:, :, 2] - t0
life = end.max(axis=1) - start.min(axis=1)
stage = (staged - start).max(axis=1)
items = life - stage
marks = w[:, 0, 3:6] - t0                                   # argmax known, nodes staged, running maximum done
phases = {"argmax_us": float((marks[:, 0] - start[:, 0]).mean()), "nodes_us": float((marks[:, 1] - marks[:, 0]).mean()),
          "running_max_us": float((marks[:, 2] - marks[:, 1]).mean()),
          "candidates_us": float((staged[:, 0] - marks[:, 2]).mean())}
print(json.dumps({"blocks": int(w.shape[0]), "kernel_us": float(end.max()),
                  "block_life_us": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "p90": float(np.percentile(life, 90))},
                  "staging_phases": phases, "staging_us": {"mean": float(stage.mean()), "p10": float(np.percentile(stage, 10)), "p90": float(np.percentile(stage, 90))},
                  "item_loop_us": {"mean": float(items.mean())},
                  "wave_slot_fill_in_item_loop": float(((end - staged).sum()) / (8 * items.sum())),
                  "sum_of_block_lives_over_kernel_x_slots": float(life.sum() / (end.max() * 512))}))
wg_start, wg_end = start.min(axis=1), end.max(axis=1)
edges = np.linspace(0, wg_end.max(), 41)
print("resident workgroups per 1/40 of the launch:", [round((np.minimum(wg_end, b) - np.maximum(wg_start, a)).clip(0).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])])
print("busy waves per 1/40 of the launch:", [round((np.minimum(end, b) - np.maximum(start, a)).clip(0).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])])
