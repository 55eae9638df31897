#!/usr/bin/env python3
"""A/B of the short-grid kernel against the general kernel on the same box (context options short_kernel,
short_queue, short_concurrent; one child process each): kernel time of BASELINE config 3 and of the
config-5 shard, and how far the two outputs are apart.  Usage: python tools/ab_short.py [n_prof]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def child(tag, n_prof):
    import torch
    from pyrayhf_amd import library, synth, _native, dist as pdist
    dev = torch.device("cuda", 0)
    ctx = _native.context(0)
    for item in os.environ["AB_OPTIONS"].split(","):
        ctx.set_option(item.split("=")[0], float(item.split("=")[1]))
    f174 = synth.sounder_frequencies(1)
    alt, den, bmag, bpsi = synth.chapman_profiles(n_prof, 20260003)
    t = [torch.as_tensor(x, device=dev) for x in (f174, den, bmag, bpsi, alt)]
    res = {"tag": tag}
    for n_points in (200, 500, 1000):
        ms = []
        for r in range(12):
            out = library.vertical_forward_operator(*t, "O", n_points, sync=True)
            if r >= 2:
                ms.append(ctx.last_kernel_ms())
        res[f"config3_n{n_points}_ms"] = float(np.median(ms))
        np.save(os.path.join(ROOT, "gpurun_out", f"ab_short_{tag}_n{n_points}.npy"), out.cpu().numpy())
    for mode, n_points, rows in (("X", 200, 10000), ("X", 500, 10000), ("X", 1000, 10000), ("X", 2000, 4000)):
        tx = [t[0]] + [x[:rows] for x in t[1:4]] + [t[4]]
        ms = []
        for r in range(8):
            out = library.vertical_forward_operator(*tx, mode, n_points, sync=True)
            if r >= 2:
                ms.append(ctx.last_kernel_ms())
        res[f"{mode}{n_points}_{rows}x174_ms"] = float(np.median(ms))
        np.save(os.path.join(ROOT, "gpurun_out", f"ab_short_{tag}_x{n_points}.npy"), out.cpu().numpy())
    from bench import CONFIG5_SEGMENTS
    rows, segs = pdist.shard_segments(CONFIG5_SEGMENTS, 8, 0)
    alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=rows)
    tt = [torch.as_tensor(x, device=dev) for x in (synth.sounder_frequencies(5), den, bmag, bpsi, alt)]
    ms = []
    for r in range(6):
        out = library.vertical_forward_operator_mixed(*tt, segs)
        torch.cuda.synchronize()
        if r >= 2:
            ms.append(ctx.last_kernel_ms())
    res["config5_shard_ms"] = float(np.median(ms))
    np.save(os.path.join(ROOT, "gpurun_out", f"ab_short_{tag}_c5.npy"), out.cpu().numpy())
    print(json.dumps(res), flush=True)

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    child(sys.argv[2], int(sys.argv[3]))
    sys.exit(0)
n_prof = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for tag, knob, qfix, conc in (("general", "0", "0", "1"), ("short", "1", "0", "1"), ("tinyq", "1", "40", "1"),
                              ("sequential", "1", "0", "0")):
    env = dict(os.environ, AB_OPTIONS=f"short_kernel={knob},shortx_kernel={knob},short_queue={qfix},short_concurrent={conc}")
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag, str(n_prof)], env=env, check=True)
for key in ("n200", "n500", "n1000", "x200", "x2000", "c5"):
    b = np.load(os.path.join(ROOT, "gpurun_out", f"ab_short_short_{key}.npy"))
    t = np.load(os.path.join(ROOT, "gpurun_out", f"ab_short_tinyq_{key}.npy"))
    print(json.dumps({"compare": key, "queue of 40 entries bit-identical to the default": bool(np.array_equal(b, t, equal_nan=True))}))
    a = np.load(os.path.join(ROOT, "gpurun_out", f"ab_short_general_{key}.npy"))
    same_mask = bool(np.array_equal(np.isnan(a), np.isnan(b)))
    ok = np.isfinite(a) & np.isfinite(b)
    rel = np.abs(a[ok] - b[ok]) / np.abs(a[ok])
    print(json.dumps({"compare": key, "nan_masks_equal": same_mask, "finite": int(ok.sum()),
                      "max_rel_diff": float(rel.max(initial=0)), "p99_rel_diff": float(np.quantile(rel, 0.99)) if rel.size else 0.0,
                      "beyond_1e-7": int((rel > 1e-7).sum()), "bit_identical": int((rel == 0).sum())}), flush=True)
