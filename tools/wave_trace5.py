"""Timeline of the config-5 per-GPU shard (6 250 x 512, mixed O/X x {200, 2000, 20000}, one work-list launch) from
a -DPRHF_TRACE build (PRHF_LIB points at it): how full the workgroup slots and wave slots are over the launch."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
path = "/tmp/prhf_trace5.bin"
os.environ["PRHF_TRACE_FILE"] = path
import bench
from pyrayhf_amd import library, synth, dist as pdist
dev = torch.device("cuda", 0)
segs = bench.config5_segments(1)
rows, local = pdist.shard_segments(segs, 1, 0)
alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=rows)
t = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(5), den, bmag, bpsi, alt)]
for _ in range(2):
    library.vertical_forward_operator_mixed(*t, local)
w = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 6).astype(np.float64) / 100.0
w = w[w[:, :, 1].max(axis=1) > 0]
t0 = w[:, :, 0].min()
start, end = w[:, :, 0] - t0, w[:, :, 1] - t0
wg_start, wg_end = start.min(axis=1), end.max(axis=1)
life = wg_end - wg_start
kernel = wg_end.max()
print(json.dumps({"blocks": int(w.shape[0]), "kernel_us": float(kernel), "segments": local,
                  "block_life_us": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "p50": float(np.median(life)),
                                    "p90": float(np.percentile(life, 90)), "max": float(life.max())},
                  "workgroup_slot_fill": float(life.sum() / (kernel * 512)),
                  "wave_slot_fill": float((end - start).sum() / (kernel * 4096))}))
edges = np.linspace(0, kernel, 41)
res = [(np.minimum(wg_end, b) - np.maximum(wg_start, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident workgroups per 1/40 of the launch:", [round(x) for x in res])
waves = [(np.minimum(end, b) - np.maximum(start, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("busy waves per 1/40 of the launch:", [round(x) for x in waves])
order = np.argsort(-life)[:12]
keep = np.flatnonzero(np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 6)[:, :, 1].max(axis=1) > 0)
print("longest blocks (block index, start us, life us):", [(int(keep[i]), round(float(wg_start[i])), round(float(life[i]))) for i in order])
by_bid = {int(keep[i]): (round(float(wg_start[i])), round(float(life[i]))) for i in range(len(keep))}
print("blocks 0..1200 every 24th (index: start us, life us):", {b: by_bid.get(b) for b in range(0, 1200, 24)})
