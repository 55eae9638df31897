#!/usr/bin/env python3
"""Benchmark of the vertical-ionogram forward operator on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload config4|config3|config5]

A "step" is one pass of the hot path over one device-resident batch of synthetic profiles:
one fused-kernel launch per GPU (plus, for N > 1, the gather of the result rows over RCCL).

Workloads (DESIGN.md "Measurement"; always the rows rank r would own at N = 8, so N = 8 is the
BASELINE configuration itself and smaller N are weak-scaling prefixes of it):
  config4 (default)  per-GPU shard of BASELINE.json configs[3] - the configuration its target
                     "integrals/s at n_points=20000 X-mode" is quoted on: 12 500 synthetic Chapman
                     profiles x 256 frequencies, X mode, n_points = 20000 per GPU;
  config3            configs[2]: 10 000 profiles x 174 frequencies, O mode, n_points = 200 (per GPU);
  config5            per-GPU shard of configs[4]: 6 250 profiles x 512 frequencies, mixed O/X x
                     n_points in {200, 2000, 20000}, ONE work-list launch per GPU.
The single-profile configs[1] (latency-bound: 174 pairs per launch) is timed too and reported under
"single_profile".

Launch: `python bench.py --gpus N` starts its own N ranks (one process per GPU, torch.distributed.run
on 127.0.0.1) when it is not already running under a launcher; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses the ranks it is given.
Rank 0 prints ONE JSON line.  `value` counts every submitted (profile, frequency) pair; the
reflecting fraction is reported beside it.  Inputs are resident in HBM before the timed region.
`roofline.traffic` (HBM bytes per launch) of the default single-GPU workload is measured by two short child
runs of this script under `rocprofv3 --pmc` before the timed run (measure_traffic_live; --no-traffic: replayed
from profiles/hbm_traffic.json); `roofline.traffic_source` says which.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 = 1/2 of the guide's 157.3 TFLOP/s vector FP32
FLOPS_PER_POINT = 68           # CSE'd nominal FP64 operations per grid point
FLOPS_PER_LEVEL = 6            # reflection search, per bottomside level and pair

CONFIG5_SEGMENTS = [(0, 20000, "O", 200), (20000, 35000, "X", 2000), (35000, 45000, "O", 2000),
                    (45000, 50000, "X", 20000)]      # BASELINE configs[4], SURVEY.md 8d


def algorithmic_bytes(n_prof, n_alt, n_freq, grid_points, per_profile_alt=False):
    """Compulsory HBM traffic of one launch; grid_points = sum of the launch's n_points."""
    return 8 * (3 * n_prof * n_alt + n_alt * (n_prof if per_profile_alt else 1) + n_freq + grid_points
                + n_prof * n_freq)


def integrated_pairs(vh, alt):
    """Pairs whose trace is an integral over the grid: a finite virtual height above the bottom of the profile.
    (Where the cutoff is exceeded at the bottom level already - X mode below the gyrofrequency - the reference's
    grid collapses onto that level and the answer is min(alt) + 1e-14 km: finite, but no work, and not counted.)"""
    return np.isfinite(vh) & (vh - float(np.min(alt)) > 1e-9)


def algorithmic_flops(vh, den, n_points_per_row, alt):
    """68 * n_points per reflecting pair + 6 * K per pair."""
    k = np.argmax(den, axis=1).astype(np.float64)
    reflecting = integrated_pairs(vh, alt).sum(axis=1).astype(np.float64)
    return float((FLOPS_PER_POINT * np.asarray(n_points_per_row, dtype=np.float64) * reflecting).sum()
                 + (FLOPS_PER_LEVEL * k * vh.shape[1]).sum())


# ------------------------------------------------------------------------------------------------
# CPU baselines (the oracle = the reference's algorithm; reported beside the GPU number, not a target)
# ------------------------------------------------------------------------------------------------
def usable_cores():
    """Host cores this process may use: the affinity mask, capped by the cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_worker(job):
    """One worker process of the all-cores leg: NumPy oracle on its own rows until the budget is spent."""
    freq, alt, den, bmag, bpsi, mode, n_points, budget_s = job
    from oracle import vfo_numpy as orc
    done, t0 = 0, time.perf_counter()
    while done < den.shape[0]:
        orc.virtual_heights(freq, den[done], bmag[done], bpsi[done], alt, mode, n_points)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(freq, alt, den, bmag, bpsi, mode, n_points, what, budget_s=10.0):
    """The oracle (NumPy restatement of the reference's CPU path), one process, on a bounded sample."""
    from oracle import vfo_numpy as orc
    orc.virtual_heights(freq, den[0], bmag[0], bpsi[0], alt, mode, min(n_points, 200))   # import / warm-up
    done, dt = _cpu_worker((freq, alt, den, bmag, bpsi, mode, n_points, budget_s))
    return {"value": done * freq.size / dt, "unit": "integrals/s", "cores": 1, "kind": "port",
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(),
            "sample": f"{done} profiles x {freq.size} freqs of {what}, {mode}-mode n_points={n_points}, "
                      f"oracle/vfo_numpy.py single process, {dt:.1f} s"}


def cpu_baseline_all_cores(freq, alt, den, bmag, bpsi, mode, n_points, what, budget_s=8.0):
    """the same oracle under multiprocessing over profiles on every usable host core."""
    import multiprocessing as mp
    cores = usable_cores()
    per = max(1, den.shape[0] // cores)
    jobs = [(freq, alt, den[i * per:(i + 1) * per], bmag[i * per:(i + 1) * per], bpsi[i * per:(i + 1) * per], mode,
             n_points, budget_s) for i in range(cores) if den[i * per:(i + 1) * per].shape[0]]
    # spawn: the workers must not inherit this process's GPU state; they import numpy and oracle/ only
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        pool.map(_cpu_worker, [(freq, alt, den[:1], bmag[:1], bpsi[:1], mode, min(n_points, 200), 0.0)] * len(jobs))
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, jobs, chunksize=1)
        wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    return {"value": done * freq.size / wall, "unit": "integrals/s", "cores": len(jobs), "kind": "port",
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(),
            "sample": f"{done} profiles x {freq.size} freqs of {what}, {mode}-mode n_points={n_points}, "
                      f"oracle/vfo_numpy.py in {len(jobs)} processes (multiprocessing, one per usable core), {wall:.1f} s"}


def cpu_baseline_c(freq, alt, den, bmag, bpsi, mode, n_points, what):
    """The fused plain-C restatement (oracle/vfo_oracle.c) on every usable core: a stronger CPU baseline than
    the reference's unfused NumPy path.  None when the library is not built."""
    from oracle import vfo_c
    if not vfo_c.available():
        return None
    cores = min(vfo_c.threads(), usable_cores())
    vfo_c.virtual_heights_batch(freq, den[:cores], bmag[:cores], bpsi[:cores], alt, mode, 200, n_threads=cores)
    n = min(den.shape[0], 4 * cores)
    t0 = time.perf_counter()
    vfo_c.virtual_heights_batch(freq, den[:n], bmag[:n], bpsi[:n], alt, mode, n_points, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n * freq.size / dt, "unit": "integrals/s", "cores": cores, "kind": "port",
            "sample": f"{n} profiles x {freq.size} freqs of {what}, {mode}-mode n_points={n_points}, "
                      f"oracle/vfo_oracle.c OpenMP x{cores}, {dt:.1f} s"}


def config5_segments(world, profiles_per_gpu=None):
    """The global work list of an N-GPU run: the first N/8 of every slice of BASELINE config 5 (N = 8: all of
    it), optionally scaled to `profiles_per_gpu` rows per GPU in the same proportions (rehearsals)."""
    segs = [(p0, p0 + ((p1 - p0) * world) // 8, m, n) for (p0, p1, m, n) in CONFIG5_SEGMENTS]
    if profiles_per_gpu:
        f = profiles_per_gpu * world / sum(p1 - p0 for p0, p1, _, _ in segs)
        segs = [(p0, p0 + max(1, int((p1 - p0) * f)), m, n) for (p0, p1, m, n) in segs]
    return segs


# ------------------------------------------------------------------------------------------------
# launcher
# ------------------------------------------------------------------------------------------------
def extra_legs(torch, dev, ctx, library, synth, pdist, math):
    """config3, config5_shard, config4_full: each {ms, integrals_per_s, roofline, ...} (see main)."""
    prof_json = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    traffic_rec = json.load(open(prof_json)) if os.path.exists(prof_json) else {}
    legs = {}

    def leg(name, what, freq, alt, den, bmag, bpsi, segments, steps, warmup, key):
        """segments: [(p0, p1, mode, n_points)] over the rows of den; one launch per step.  Host arrays or tensors
        on `dev`; the nominal flop count (algorithmic_flops) is taken on the device."""
        tt = [torch.as_tensor(x, device=dev) for x in (freq, den, bmag, bpsi, alt)]
        single = len(segments) == 1
        mode, n_points = segments[0][2], segments[0][3]
        outs = None
        for i in range(warmup + steps):
            if i == warmup:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            if single:
                outs = library.vertical_forward_operator(*tt, mode, n_points, math=math, sync=False, out=outs)
            else:
                outs = library.vertical_forward_operator_mixed(*tt, segments, math=math, sync=False)
        torch.cuda.synchronize(dev)
        wall_ms = 1e3 * (time.perf_counter() - t0) / steps
        k_ms = float(np.mean(ctx.recent_kernel_ms(steps)))
        n_prof, n_freq, n_alt = int(tt[1].shape[0]), int(tt[0].numel()), int(tt[4].numel())
        n_points_row = torch.as_tensor(np.concatenate([np.full(p1 - p0, n) for p0, p1, _, n in segments]),
                                       dtype=torch.float64, device=dev)
        integrated = torch.isfinite(outs) & (outs - tt[4].min() > 1e-9)          # integrated_pairs()
        aflops = float((FLOPS_PER_POINT * n_points_row * integrated.sum(dim=1)).sum()
                       + (FLOPS_PER_LEVEL * torch.argmax(tt[1], dim=1).double() * n_freq).sum())
        abytes = algorithmic_bytes(n_prof, n_alt, n_freq, sum(n for _, _, _, n in segments))
        rec = traffic_rec.get(key, {})
        pairs = n_prof * n_freq
        legs[name] = {
            "workload": what, "steps": steps, "ms": k_ms, "ms_per_step_wall": wall_ms,
            "integrals_per_s": pairs / (k_ms * 1e-3), "pairs_per_step": pairs,
            "reflecting_fraction": float(integrated.double().mean()),
            "roofline": {"bound": "fp64_valu", "achieved": aflops / (k_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": aflops / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "traffic": rec.get("hbm_bytes_per_launch"), "traffic_key": key if rec else None,
                         "traffic_source": (f"NOT measured in this run: replayed from profiles/hbm_traffic.json[{key}] "
                                            f"({rec.get('source')})") if rec else None,
                         "algorithmic_bytes": abytes},
        }
        del tt, outs, integrated
        torch.cuda.empty_cache()

    alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003)
    f3 = synth.sounder_frequencies(3)
    leg("config3", "BASELINE configs[2]: 10000 synthetic Chapman profiles x 174 freqs, O-mode, n_points=200, seed 20260003",
        f3, alt, den, bmag, bpsi, [(0, 10000, "O", 200)], 20, 3, "O_200_10000x174")
    rows, segs = pdist.shard_segments(CONFIG5_SEGMENTS, 8, 0)
    alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=rows)
    f5 = synth.sounder_frequencies(5)
    leg("config5_shard", "BASELINE configs[4] per-GPU shard: " + ", ".join(f"{p1 - p0} x {m}/{n}" for p0, p1, m, n in segs) +
        f" x {f5.size} freqs, ONE work-list launch, seed 20260005", f5, alt, den, bmag, bpsi, segs, 10, 2,
        f"mixed_0_{rows.size}x{f5.size}")
    alt, den, bmag, bpsi = synth.chapman_profiles_torch(100000, 20260004, dev)       # (NumPy would take ~10 s on the host)
    f4 = synth.sounder_frequencies(4)
    leg("config4_full", "BASELINE configs[3] in full on ONE GPU: 100000 synthetic Chapman profiles x 256 freqs, X-mode, "
        "n_points=20000, one launch, seed 20260004 (profiles built on the device)", f4, alt, den, bmag, bpsi, [(0, 100000, "X", 20000)], 3, 1,
        "X_20000_100000x256")
    return legs


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0    # wave instructions per second: 1024 SIMDs, one FP64-rate VALU instruction per 4
                                           # cycles, 2.4 GHz (MI355X_MICROARCH.md: 256 CUs, max clock 2400 MHz)


def f_row_legs(torch, dev, ctx, library, synth):
    """The rows SURVEY 8(f) marks "next", each with the bound it is priced against:
      snell_per_ray / snell_fan   reference trace_ray_cartesian_snells / trace_ray_spherical_snells (library.py:1096-1268,
                                  :1460-1713): rays/s; issue-bound - vector instructions per ray (from the committed profile,
                                  profiles/f_row_bounds.json) x rays/s against the chip's FP64-rate issue slots
      stage_ops                   find_mu_mup (:161-256), find_vh (:259-293), regrid_to_nonuniform_grid (:324-438) as
                                  device ops on 2^26 elements: algorithmic bytes / kernel time against 8 TB/s
      fit_brute                   the brute-force search of minimize_parameters (:794-798): 41 x 41 candidate profiles x 174
                                  frequencies through prhf_vfo_residual_f64; nominal FP64 flops like config 3
    Kernel times are the library's HIP events; about 2 s of wall time in all."""
    from pyrayhf_amd import _native, fitting, tracers
    bounds_path = os.path.join(ROOT, "profiles", "f_row_bounds.json")
    bounds = json.load(open(bounds_path)) if os.path.exists(bounds_path) else {}
    legs = {}

    def issue_roofline(key, rays_per_s):
        b = bounds.get(key)
        if not b:
            return {"bound": "valu_issue", "achieved": None, "peak": VALU_ISSUE_PEAK, "unit": "wave-instructions/s",
                    "frac": None, "traffic": None, "note": "no committed instruction count for this kernel yet"}
        achieved = b["valu_per_ray"] * rays_per_s
        return {"bound": "valu_issue", "achieved": achieved, "peak": VALU_ISSUE_PEAK, "unit": "wave-instructions/s",
                "frac": achieved / VALU_ISSUE_PEAK, "valu_per_ray": b["valu_per_ray"], "traffic": b.get("hbm_bytes_per_launch"),
                "traffic_source": f"NOT measured in this run: {b.get('source')}",
                "note": "vector instructions per ray (SQ_INSTS_VALU / rays of the committed profile) x rays/s against 1024 SIMDs "
                        "x 2.4 GHz / 4 cycles per FP64-rate instruction; v_rsq_f64 / v_rcp_f64 issue at a quarter of that rate"}

    # ---- Snell's-law tracers: 200 000 random (frequency, elevation, profile) rays over 256 profiles, O mode -------------
    alt, den, bmag, bpsi = synth.chapman_profiles(256, 7)
    rng = np.random.default_rng(0)
    n_rays = 200000
    f = rng.uniform(2e6, 14e6, n_rays)
    e = rng.uniform(5.0, 89.0, n_rays)
    idx = rng.integers(0, 256, n_rays)
    per_ray = {"workload": "200000 rays, random frequency 2-14 MHz, elevation 5-89 deg, one of 256 synthetic Chapman profiles "
                           "(620 levels) each, O mode, one launch (host arrays in and out; kernel time from HIP events)"}
    for name, fn in (("flat", tracers.trace_rays_cartesian_snells), ("spherical", tracers.trace_rays_spherical_snells)):
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = fn(f, e, alt, den, bmag, bpsi, "O", profile_index=idx)
            wall = time.perf_counter() - t0
            ms.append(ctx.last_kernel_ms())
        k = min(ms[1:])
        per_ray[name] = {"kernel_ms": k, "rays_per_s": n_rays / (k * 1e-3), "ms_per_call_wall": 1e3 * wall,
                         "traced_fraction": float(np.isfinite(r["group_path_km"]).mean()),
                         "roofline": issue_roofline(f"snell_per_ray_{name}", n_rays / (k * 1e-3))}
    legs["snell_per_ray"] = per_ray
    # ---- fans: 16 profiles x 100 frequencies x 128 elevations, the levels once per (profile, frequency) ------------------
    fan_f, fan_e = np.linspace(2e6, 14e6, 100), np.linspace(5.0, 89.0, 128)
    fan = {"workload": "16 profiles x 100 frequencies x 128 elevations = 204800 rays, O mode, grouped launch (refractive-index "
                       "levels once per (profile, frequency))"}
    for name, fn in (("flat", tracers.trace_fan_cartesian_snells), ("spherical", tracers.trace_fan_spherical_snells)):
        ms = []
        for _ in range(3):
            r = fn(fan_f, fan_e, alt, den[:16], bmag[:16], bpsi[:16], "O")
            ms.append(ctx.last_kernel_ms())
        k, n_fan = min(ms[1:]), 16 * 100 * 128
        fan[name] = {"kernel_ms": k, "rays_per_s": n_fan / (k * 1e-3),
                     "traced_fraction": float(np.isfinite(r["group_path_km"]).mean()),
                     "roofline": issue_roofline(f"snell_fan_{name}", n_fan / (k * 1e-3))}
    legs["snell_fan"] = fan

    # ---- stage ops on GPU-resident arrays (HBM-bound) ---------------------------------------------------------------------
    DP = _native.FLAG_DEVICE_PTRS

    def best(call, reps=4):
        ms = []
        torch.cuda.synchronize(dev)        # the operands were made on torch's stream; the ops run on the context's own
        for _ in range(reps):
            _native.raise_for(call())
            ms.append(ctx.last_kernel_ms())
        return min(ms[1:])

    def hbm(bytes_moved, ms):
        gbs = bytes_moved / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": None, "algorithmic_bytes": bytes_moved}
    ops = {}
    n = 1 << 26
    X = torch.rand(n, dtype=torch.float64, device=dev) * 0.9
    Y = torch.rand(n, dtype=torch.float64, device=dev) * 0.3 + 0.05
    P = torch.rand(n, dtype=torch.float64, device=dev) * 90.0
    mu, mup = torch.empty_like(X), torch.empty_like(X)
    ctx.set_stream(0, borrow=False)
    ctx.set_math(_native.MATH_FAST)
    ms = best(lambda: ctx.mu_mup(X.data_ptr(), Y.data_ptr(), P.data_ptr(), n, 1, mu.data_ptr(), mup.data_ptr(), DP))
    ops["find_mu_mup"] = {"workload": f"{n} elements, X mode, fast tier: 24 B read + 16 B written per element", "kernel_ms": ms,
                          "elements_per_s": n / (ms * 1e-3), "roofline": hbm(40 * n, ms)}
    del mu, mup
    F, N = 4096, 16384                                   # 2^26 grid points
    Xr, Yr, Pr = X.view(F, N), Y.view(F, N), P.view(F, N)
    Dr = torch.rand(F, N, dtype=torch.float64, device=dev) * 0.01
    vh = torch.empty(F, dtype=torch.float64, device=dev)
    ms = best(lambda: ctx.find_vh(Xr.data_ptr(), Yr.data_ptr(), Pr.data_ptr(), Dr.data_ptr(), F, N, 80.0, 1, vh.data_ptr(), DP))
    ops["find_vh"] = {"workload": f"({F}, {N}) arrays X, Y, psi, dh -> ({F},), X mode, fast tier: 32 B read per grid point",
                      "kernel_ms": ms, "points_per_s": F * N / (ms * 1e-3), "roofline": hbm(32 * F * N, ms)}
    del X, Y, P, Xr, Yr, Pr, Dr
    torch.cuda.empty_cache()
    a1, d1, b1, p1 = synth.chapman_profiles(1, 5)
    F, N = 1024, 20000
    fhz = torch.as_tensor(np.linspace(0.5, 16.0, F) * 1e6, device=dev)
    tt = [torch.as_tensor(x, device=dev) for x in (d1[0], b1[0], p1[0], a1)]
    mult = torch.as_tensor(np.array(library._multiplier(N)), device=dev)
    outs = [torch.empty(F, N, dtype=torch.float64, device=dev) for _ in range(7)] + [torch.empty(F, N, dtype=torch.int64, device=dev)]
    ptrs = [o.data_ptr() for o in outs]
    ms = best(lambda: ctx.regrid(fhz.data_ptr(), F, tt[0].data_ptr(), tt[1].data_ptr(), tt[2].data_ptr(), tt[3].data_ptr(),
                                 a1.size, mult.data_ptr(), N, 1, ptrs, DP))
    ops["regrid_to_nonuniform_grid"] = {"workload": f"1 profile x {F} frequencies x {N} points, X mode: eight (F, N) arrays written, "
                                                    "64 B per grid point", "kernel_ms": ms, "points_per_s": F * N / (ms * 1e-3),
                                        "roofline": hbm(64 * F * N, ms)}
    del outs
    torch.cuda.empty_cache()
    ctx.set_math(_native.MATH_AUTO)
    legs["stage_ops"] = ops

    # ---- the brute-force search of minimize_parameters: a 41 x 41 grid of candidate layers, one launch ------------------
    alt1 = np.arange(80.0, 700.0, 1.0)
    hm = np.linspace(250.0, 350.0, 41)
    hf = np.linspace(35.0, 70.0, 41)

    def chapman(nm, h0, hs):
        z = (alt1[None, :] - h0[:, None]) / hs[:, None]
        return nm * np.exp(0.5 * (1.0 - z - np.exp(-z)))
    hh, ss = (g.ravel() for g in np.meshgrid(hm, hf, indexing="ij"))
    cand = chapman(8e11, hh, ss) + chapman(1.2e11, np.full(hh.size, 110.0), np.full(hh.size, 9.0))
    bmag1 = 4.5e-5 * ((6371.0 + 80.0) / (6371.0 + alt1)) ** 3
    bpsi1 = 40.0 + 0.001 * (alt1 - 80.0)
    f1 = synth.sounder_frequencies(1)
    true = cand[20 * 41 + 20]
    vh_obs = library.vertical_forward_operator(f1, true, bmag1, bpsi1, alt1, "O", 200)
    keep = np.isfinite(vh_obs)
    fk, ok_ = f1[keep], vh_obs[keep]
    tc = [torch.as_tensor(x, device=dev) for x in (fk, ok_, cand, bmag1, bpsi1, alt1)]
    ms, walls = [], []
    for _ in range(4):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        _res, cost, vhm = fitting.residual_VH_batch(*tc, "O", 200, return_vh=True)
        torch.cuda.synchronize(dev)
        walls.append(time.perf_counter() - t0)
        ms.append(ctx.last_kernel_ms())
    best_node = int(torch.argmin(torch.nan_to_num(cost, nan=float("inf"))).item())
    k = min(ms[1:])
    integrated = torch.isfinite(vhm) & (vhm - float(alt1.min()) > 1e-9)
    aflops = float(FLOPS_PER_POINT * 200 * integrated.sum() + (FLOPS_PER_LEVEL * torch.argmax(tc[2], dim=1).double() * fk.size).sum())
    legs["fit_brute"] = {"workload": f"41 x 41 = {cand.shape[0]} candidate Chapman layers (hmF2 x scale height) x {fk.size} reflecting "
                                     "frequencies of a 174-frequency sweep, O mode, n_points=200, shared field: ONE call of "
                                     "prhf_vfo_residual_f64 (fused operator + residual rows + sums of squares), GPU-resident",
                         "kernel_ms": k, "ms_per_call_wall": 1e3 * min(walls[1:]), "candidates_per_s": cand.shape[0] / (k * 1e-3),
                         "integrals_per_s": cand.shape[0] * fk.size / (k * 1e-3), "best_node_is_the_generating_one": best_node == 20 * 41 + 20,
                         "roofline": {"bound": "fp64_valu", "achieved": aflops / (k * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": aflops / (k * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, "traffic": None,
                                      "note": "HIP events around the whole call: the fused operator over the candidates and the "
                                              "residual kernel behind it; 1 681 blocks are 1.6 resident rounds of the short-grid "
                                              "kernel: a launch this small is drain-bound"}}
    return legs


def host_buffer_leg(library, freq, alt, den, bmag, bpsi, mode, n_points, math, kernel_ms):
    """A batch handed over as host NumPy buffers (pageable): H2D + kernel + D2H, wall time of the second call (the first
    grows the library's staging arena).  PCIe-inclusive - never `value`."""
    library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points, math=math)
    walls = []
    for _ in range(3):
        t2 = time.perf_counter()
        library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points, math=math)
        walls.append(time.perf_counter() - t2)
    dt = min(walls)
    n_prof, n_freq = den.shape[0], freq.size
    bytes_in = 8 * (3 * den.size + alt.size + freq.size)
    bytes_out = 8 * n_prof * n_freq
    return {"integrals_per_s": n_prof * n_freq / dt, "ms_per_call": 1e3 * dt, "h2d_bytes": bytes_in, "d2h_bytes": bytes_out,
            "effective_h2d_gbs": bytes_in / dt / 1e9, "kernel_ms_resident": kernel_ms,
            "ratio_to_resident_kernel": (1e3 * dt / kernel_ms) if kernel_ms else None,
            "note": "PCIe-inclusive (pageable host memory in and out); never `value`; effective_h2d_gbs = input bytes / the whole "
                    "call (upload, kernel and download overlap in slabs: where the kernel dominates - config 4 - it says nothing "
                    "about the link)"}


def devices_all_child(n_visible):
    """(child process of the default run) config 4's rows, 12 500 per visible GPU, through
    vertical_forward_operator(..., devices="all") on NumPy arrays; prints one JSON object."""
    import torch

    from pyrayhf_amd import library, synth
    dev = torch.device("cuda", 0)
    rows_all = 12500 * n_visible
    a_t, d_t, b_t, p_t = synth.chapman_profiles_torch(max(rows_all, 100000), 20260004, dev, rows=slice(0, rows_all))
    host = [x.cpu().numpy() for x in (d_t, b_t, p_t)]
    alt = a_t.cpu().numpy()
    del d_t, b_t, p_t
    f4 = synth.sounder_frequencies(4)
    library.vertical_forward_operator(f4, *host, alt, "X", 20000, devices="all")         # (arenas grow)
    t3 = time.perf_counter()
    out = library.vertical_forward_operator(f4, *host, alt, "X", 20000, devices="all")
    dt3 = time.perf_counter() - t3
    print(json.dumps({"gpus": n_visible, "profiles": rows_all, "ms_per_call": 1e3 * dt3,
                      "integrals_per_s": rows_all * f4.size / dt3, "finite_fraction": float(np.isfinite(out).mean()),
                      "note": "vertical_forward_operator(..., devices='all') on NumPy arrays: in-process threads, one per GPU; "
                              "PCIe-inclusive; never `value`"}))


def devices_all_leg(n_visible, limit_s=150):
    import signal
    cmd = [sys.executable, os.path.abspath(__file__), "--devices-all-child", str(n_visible)]
    try:
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, start_new_session=True)
        try:
            out, _ = child.communicate(timeout=limit_s)
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, signal.SIGKILL)
            child.wait()
            return {"error": f"no answer within {limit_s} s (child ended)"}
        if child.returncode != 0:
            return {"error": f"child exit code {child.returncode}"}
        return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    except Exception as exc:       # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def arm_watchdog(seconds, what, rank):
    """A hung collective set-up must end with a message and a non-zero exit, not with the driver's time limit: after
    `seconds` the process says what it was waiting for and leaves (os._exit: no exec, no re-launch)."""
    import threading

    def fire():
        sys.stderr.write(f"bench.py rank {rank}: {what} did not finish within {seconds} s - giving up (exit 4)\n")
        sys.stderr.flush()
        os._exit(4)
    timer = threading.Timer(seconds, fire)
    timer.daemon = True
    timer.start()
    return timer


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) and wait.

    This parent never touches the GPU (no torch.cuda call, no libprhf): the ranks are fresh child
    processes of `python -m torch.distributed.run`, whose stdout (rank 0's JSON line) passes through."""
    port = free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes on this image)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def measure_traffic_live(kernel_substring="vfo_kernel<1"):
    """HBM bytes per launch of the default workload's kernel, measured NOW: two short child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE - counters in their own passes, never beside a trace), started
    before this process touches the GPU.  Units and the gfx950 correction as MI355X_MICROARCH.md prescribes (both
    counters in KiB; FETCH_SIZE tallies a wide coalesced read at half its bytes: doubled).  Returns a dict, or None
    when anything at all goes wrong - the caller then falls back to the committed profile's figure and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    if not os.path.exists("/dev/kfd") or shutil.which("rocprofv3") is None:
        return None
    # (this run is itself under a profiler - tools/profile.sh, a judge's own rocprofv3 pass: no profiler inside a profiler)
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="prhf_traffic_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-single-profile",
                   "--no-legs", "--no-traffic"]
            env = dict(os.environ, TMPDIR="/tmp", PRHF_BENCH_CHILD="1")
            # (a process group of its own: a pass that overruns is ended with everything it started, so that nothing of
            #  it is still on the GPU when the timed run begins)
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)
            try:
                code = child.wait(timeout=240)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)
                child.wait()
                return None
            if code != 0:
                return None
            values = []
            for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                with open(path) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") == ctr and kernel_substring in r.get("Kernel_Name", ""):
                            values.append(float(r["Counter_Value"]))
            if not values:
                return None
            out[ctr] = sum(values) / len(values)
            out[ctr + "_dispatches"] = len(values)
        out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
        return out
    except Exception:                          # noqa: BLE001 - a profiler that is missing, refused or slow must not sink the bench line
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config4", choices=["config4", "config3", "config5"])
    ap.add_argument("--profiles", type=int, default=None, help="profiles per GPU (default: the workload's)")
    ap.add_argument("--freqs", type=int, default=None)
    ap.add_argument("--n-points", type=int, default=None)
    ap.add_argument("--mode", default=None, choices=["O", "X"])
    ap.add_argument("--math", default=None, choices=[None, "faithful", "fast"])
    ap.add_argument("--gather", default="root", choices=["root", "all"],
                    help="N > 1: result rows gathered on rank 0 alone (dist.gather: each peer's block over its own xGMI link, "
                         "SURVEY 8e) or on every rank (all_gather_into_tensor)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-profile", action="store_true")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="a context option of the library (prhf_ctx_set_option), e.g. shortx_kernel=0: A/B measurements")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the extra driver-timed legs (config3, config5_shard, config4_full) of the default run")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure the HBM traffic of the default workload live (two short child runs under rocprofv3 "
                         "--pmc before the timed run): roofline.traffic is then replayed from profiles/hbm_traffic.json")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: initialise the process group anyway and push the result rows through the real "
                         "all_gather_into_tensor (RCCL with one rank) - the N > 1 code path on one GPU")
    ap.add_argument("--devices-all-child", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.devices_all_child:
        devices_all_child(args.devices_all_child)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))

    # roofline.traffic of the default single-GPU workload: measured in this run, by child processes, before this process
    # initialises the GPU (the PMC passes cannot share a process with the timed run)
    live_traffic = None
    default_workload = (args.workload == "config4" and args.profiles is None and args.freqs is None and
                        args.n_points is None and args.mode is None and args.math is None and not args.option)
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and default_workload and not args.no_traffic and
            not args.force_collective and not os.environ.get("PRHF_BENCH_CHILD")):
        live_traffic = measure_traffic_live()

    # stdout carries the one JSON line and nothing else: whatever the libraries below print on file descriptor 1
    # (Gloo announces its connections there) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from pyrayhf_amd import _native, library, synth
    from pyrayhf_amd import dist as pdist

    rank, world, local_rank = pdist.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists in the product path)")
    # Rehearsal knob for a one-GPU box: PRHF_BENCH_BACKEND=gloo runs every rank on GPU 0 and gathers
    # through host memory.  The driver's runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get("PRHF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    elif world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPUs are visible "
                         "(PRHF_BENCH_BACKEND=gloo rehearses N ranks on one GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collective
    world_seen = 1
    if collective:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # The first N > 1 run on hardware must fail loudly, not hang: collectives time out after 120 s (the default is
        # ten minutes - the driver's whole limit), a set-up that is still not through after 180 s ends the rank with a
        # message, and an exception in it is reported with the rank and the backend before the non-zero exit.
        limit = datetime.timedelta(seconds=120)
        watchdog = arm_watchdog(180, f"process-group set-up ({backend}: init_process_group and the first gather / all-gather, "
                                     f"world size {world})", rank)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=limit)
            else:
                dist.init_process_group(backend, timeout=limit)
            world_seen = dist.get_world_size()
            # Communicator set-up (RCCL builds its rings and its send/recv channels on first use: seconds) belongs to the
            # environment, not to a step: one tiny exchange of each kind the steps use, whatever --warmup is.
            side0 = dev if backend == "nccl" else torch.device("cpu")
            tiny = torch.zeros((1, 8), dtype=torch.float64, device=side0)
            pdist._exchange(tiny, world_seen, None, None)
            pdist._exchange(tiny, world_seen, None, 0)
            if backend == "nccl":
                torch.cuda.synchronize(dev)
            dist.barrier()
        except Exception as exc:           # noqa: BLE001
            sys.stderr.write(f"bench.py rank {rank}/{world}: process-group set-up over {backend} failed: "
                             f"{type(exc).__name__}: {exc}\n")
            sys.stderr.flush()
            os._exit(3)
        watchdog.cancel()

    math = {None: None, "faithful": _native.MATH_FAITHFUL, "fast": _native.MATH_FAST}[args.math]
    ctx = _native.context(local_rank)
    for item in args.option:
        ctx.set_option(item.split("=")[0], float(item.split("=")[1]))

    # ---- the workload: this rank's rows of the BASELINE configuration ------------------------------
    if args.workload == "config5":
        n_freq = args.freqs or 512
        freq = np.linspace(0.5, 16.0, n_freq)
        # N GPUs take the first N/8 of every slice of config 5; rank r its shard_bounds block of each
        global_segs = config5_segments(world, args.profiles)
        rows, local_segs = pdist.shard_segments(global_segs, world, rank)
        alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=rows)
        p_gpu = int(rows.size)
        p_total = sum(p1 - p0 for p0, p1, _, _ in global_segs)
        n_points_row = np.concatenate([np.full(p1 - p0, n) for p0, p1, _, n in local_segs])
        grid_points = sum(n for _, _, _, n in local_segs)
        mode, n_points = "mixed", 0
        what = "BASELINE configs[4] (50000 x 512 mixed work list)"
        workload = (f"BASELINE configs[4] per-GPU shard: {p_gpu} synthetic Chapman profiles x {n_freq} freqs (0.5-16 MHz) "
                    f"per GPU in one work-list launch, slices " +
                    ", ".join(f"{p1 - p0} x {m}/{n}" for p0, p1, m, n in local_segs) +
                    ", seed 20260005; N=8 is config 5 (50000 x 512)")
    else:
        c3 = args.workload == "config3"
        p_gpu = args.profiles or (10000 if c3 else 12500)
        n_freq = args.freqs or (174 if c3 else 256)
        n_points = args.n_points or (200 if c3 else 20000)
        mode = args.mode or ("O" if c3 else "X")
        freq = synth.sounder_frequencies(3) if (c3 and n_freq == 174) else np.linspace(0.5, 16.0, n_freq)
        p_total = p_gpu * world
        lo, hi = pdist.shard_bounds(p_total, world, rank)
        if c3:      # config 3 is a one-GPU configuration: N ranks = N independent draws of its size (replicas)
            alt, den, bmag, bpsi = synth.chapman_profiles(p_total, 20260003, rows=slice(lo, hi))
            what = "BASELINE configs[2] (10000 x 174, O/200)"
            workload = (f"BASELINE configs[2]: {p_gpu} synthetic Chapman profiles x {n_freq} freqs per GPU, {mode}-mode, "
                        f"n_points={n_points}, seed 20260003")
        else:       # always the rows of config 4's 100 000-profile draw: N = 1 is its first shard, N = 8 all of it
            alt, den, bmag, bpsi = synth.chapman_profiles(max(p_total, 100000), 20260004, rows=slice(lo, hi))
            what = "BASELINE configs[3] (100000 x 256, X/20000)"
            workload = (f"BASELINE configs[3] per-GPU shard: {p_gpu} synthetic Chapman profiles x {n_freq} freqs "
                        f"(0.5-16 MHz) per GPU, {mode}-mode, n_points={n_points}, seed 20260004; N=8 is config 4 "
                        f"(100000 x 256)")
        n_points_row = np.full(p_gpu, n_points)
        grid_points = n_points
        local_segs = global_segs = None

    t = {k: torch.as_tensor(v, device=dev) for k, v in
         (("freq", freq), ("alt", alt), ("den", den), ("bmag", bmag), ("bpsi", bpsi))}
    out = torch.empty((p_gpu, n_freq), dtype=torch.float64, device=dev)

    gather_events = []                       # (before, after) the gather of every step, on torch's current stream
    gather_dst = 0 if args.gather == "root" else None

    def step():
        nonlocal out
        if local_segs is not None:
            out = library.vertical_forward_operator_mixed(t["freq"], t["den"], t["bmag"], t["bpsi"], t["alt"],
                                                          local_segs, math=math, sync=False)
        else:
            library.vertical_forward_operator(t["freq"], t["den"], t["bmag"], t["bpsi"], t["alt"], mode, n_points,
                                              math=math, sync=False, out=out)
        if collective:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if local_segs is not None:
                pdist.gather_mixed(out, global_segs, max(p1 for _, p1, _, _ in global_segs), force=True, dst=gather_dst)
            else:
                pdist.gather_rows(out, p_total, force=True, dst=gather_dst)
            e1.record()
            gather_events.append((e0, e1))

    def fence():
        torch.cuda.synchronize(dev)
        if collective:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    gather_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                               # enqueued back to back: no host round trip inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    _native.raise_for(ctx.sync())
    # HIP events on the launch stream, recorded by the library around every launch of the timed region
    # (the context remembers the last 64)
    kernel_ms = ctx.recent_kernel_ms(min(args.steps, 64))
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else None
    kernel_ms_per_rank = None
    if collective:
        side = dev if backend == "nccl" else "cpu"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=side)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank's mean kernel time and gather time, so that a scaling loss can be attributed
        every = pdist.gather_scalars([float(np.mean(kernel_ms)), gather_ms or 0.0], device=side)
        kernel_ms_per_rank = {"min": float(every[:, 0].min()), "max": float(every[:, 0].max()),
                              "ranks": [float(v) for v in every[:, 0]]}
        gather_ms = float(every[:, 1].max())

    if rank == 0:
        vh = out.cpu().numpy()
        ms_step = 1e3 * elapsed / args.steps
        k_ms = float(np.mean(kernel_ms))
        abytes = algorithmic_bytes(p_gpu, alt.size, n_freq, grid_points)
        aflops = algorithmic_flops(vh, den, n_points_row, alt)
        traffic, traffic_src = None, None
        prof_json = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof_json):
            with open(prof_json) as fh:
                rec = json.load(fh)
            key = f"{mode}_{n_points}_{p_gpu}x{n_freq}"
            if key in rec:
                traffic = rec[key].get("hbm_bytes_per_launch")
                traffic_src = (f"NOT measured in this run: replayed from profiles/hbm_traffic.json[{key}] "
                               f"(rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE passes of this workload, {rec[key].get('source')})")
        if live_traffic is not None:
            replayed = traffic
            traffic = live_traffic["hbm_bytes_per_launch"]
            traffic_src = (f"measured in this run: two child runs of this script (2 steps each) under rocprofv3 --pmc before the "
                           f"timed run - FETCH_SIZE {live_traffic['FETCH_SIZE']:.1f} KiB x 2 (gfx950: a wide coalesced read is "
                           f"tallied at half its bytes) + WRITE_SIZE {live_traffic['WRITE_SIZE']:.1f} KiB, mean of "
                           f"{live_traffic['FETCH_SIZE_dispatches']} dispatches of the fused kernel" +
                           (f"; the committed profile (profiles/hbm_traffic.json) has {replayed:.0f} B" if replayed else ""))
        default_tier = _native.MATH_FAST if (mode == "X" and math is None) else (math if math is not None
                                                                                  else _native.MATH_FAITHFUL)
        result = {
            "metric": "virtual-height integrals/s (profile x frequency pairs)",
            "value": p_total * n_freq * args.steps / elapsed,
            "unit": "integrals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload,
                       "profiles_per_gpu": p_gpu, "n_freq": n_freq, "n_points": n_points or "200/2000/20000",
                       "mode": mode, "n_alt": int(alt.size), "math": args.math or "default",
                       "parallelism": (f"profile shards x{world}, one process per GPU, "
                                       f"{'gather of vh rows on rank 0' if args.gather == 'root' else 'all_gather of vh rows'} over "
                                       f"{'RCCL' if backend == 'nccl' else backend}") if world > 1 else "single GPU"},
            "world_size_seen": world_seen, "backend": backend if collective else None,
            "kernel_ms_per_rank": kernel_ms_per_rank,
            "gather_ms": gather_ms,          # the slowest rank's mean time in the all-gather of the result rows (null: no collective)
            "reflecting_fraction": float(integrated_pairs(vh, alt).mean()),
            "finite_fraction": float(np.isfinite(vh).mean()),
            "workgroups_per_cu": ctx.occupancy(alt.size, default_tier),
            "kernel_ms": k_ms,
            # the FP64 vector ALU (not MFMA, not HBM) binds this path, so the primary
            # roofline object prices the nominal 68 flop/point against the 78.6 TFLOP/s vector peak;
            # the compulsory-bytes view (arithmetic intensity ~1e4 flop/B) rides along as roofline_hbm.
            "roofline": {"bound": "fp64_valu", "achieved": aflops / (k_ms * 1e-3) / 1e12,
                         "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": aflops / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "flops_per_point": FLOPS_PER_POINT,
                         "note": "non-MFMA FP64 vector roofline (SURVEY 8d): nominal 68 flop/point x reflecting points "
                                 "+ 6 flop/level/pair, over the fused kernel's HIP-event time on this rank"},
            "roofline_hbm": {"bound": "hbm", "achieved": abytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": abytes,
                             "note": "compulsory bytes only; the fused kernel is not HBM bound by construction"},
        }

        if world == 1 and not args.no_single_profile:
            # BASELINE configs[1]: one profile x 174 freqs, X mode, n_points = 20000 (latency-bound)
            f1 = synth.sounder_frequencies(1)
            tf1 = torch.as_tensor(f1, device=dev)
            o1 = torch.empty((1, f1.size), dtype=torch.float64, device=dev)
            args1 = (tf1, t["den"][:1], t["bmag"][:1], t["bpsi"][:1], t["alt"], "X", 20000)
            for _ in range(3):
                library.vertical_forward_operator(*args1, math=math, sync=False, out=o1)
            torch.cuda.synchronize(dev)
            reps, t1 = 50, time.perf_counter()
            for _ in range(reps):
                library.vertical_forward_operator(*args1, math=math, sync=False, out=o1)
            torch.cuda.synchronize(dev)
            dt1 = (time.perf_counter() - t1) / reps
            result["single_profile"] = {"workload": "configs[1]: 1 profile x 174 freqs, X-mode, n_points=20000",
                                        "ms_per_call": 1e3 * dt1, "integrals_per_s": f1.size / dt1,
                                        "kernel_ms": ctx.last_kernel_ms()}

        if world == 1 and not args.no_single_profile:
            # BASELINE configs[0] and configs[1] as a PyRayHF user makes the call: one profile, NumPy arrays in,
            # a NumPy array out (wall time per call; inputs in host memory - a latency, never `value`)
            f1 = synth.sounder_frequencies(1)
            one = [np.ascontiguousarray(x) for x in (den[0], bmag[0], bpsi[0], alt)]
            legs = {}
            for name, (m1, n1) in (("config1_O_200", ("O", 200)), ("config2_X_20000", ("X", 20000))):
                for _ in range(5):
                    library.vertical_forward_operator(f1, *one, m1, n1)
                reps, t1 = 100, time.perf_counter()
                for _ in range(reps):
                    library.vertical_forward_operator(f1, *one, m1, n1)
                legs[name] = {"us_per_call": 1e6 * (time.perf_counter() - t1) / reps}
            result["dropin_call"] = dict(legs, workload="configs[0] / configs[1]: 1 profile x 174 freqs through "
                                         "vertical_forward_operator on NumPy arrays (host buffers in and out)")

        if world == 1 and not args.no_single_profile and local_segs is None:
            # the same batch handed over as host NumPy buffers (pageable): H2D + kernel + D2H
            # first call: the library's staging arena grows to this batch (hipMalloc); second call: steady state
            result["host_buffers"] = host_buffer_leg(library, freq, alt, den, bmag, bpsi, mode, n_points, math, k_ms)

        if world == 1 and not args.no_legs and args.workload == "config4" and args.profiles is None:
            # The other BASELINE configurations that fit one GPU, timed by the same process right behind the headline
            # region (kernel time from the library's HIP events; inputs resident in HBM): BASELINE configs[2], the
            # per-GPU shard of configs[4] as ONE work-list launch, and configs[3] in full (100 000 x 256) as one launch.
            del out
            for k in ("den", "bmag", "bpsi"):
                t.pop(k)
            torch.cuda.empty_cache()
            result.update(extra_legs(torch, dev, ctx, library, synth, pdist, math))
            # config 3 from NumPy arrays: 149 MB in, 14 MB out around a half-millisecond kernel - the drop-in rate of a
            # short-grid batch is PCIe's
            a3, d3, b3, p3 = synth.chapman_profiles(10000, 20260003)
            result["host_buffers_config3"] = host_buffer_leg(library, synth.sounder_frequencies(3), a3, d3, b3, p3, "O", 200, math,
                                                             result["config3"]["ms"])
            del a3, d3, b3, p3
            try:
                result.update(f_row_legs(torch, dev, ctx, library, synth))
            except Exception as exc:   # noqa: BLE001 - a failure in a row marked "next" must not sink the headline line
                result["f_row_legs_error"] = f"{type(exc).__name__}: {exc}"

        n_visible = torch.cuda.device_count()
        if world == 1 and n_visible > 1 and not args.no_legs and args.workload == "config4" and args.profiles is None:
            # More than one GPU visible to a single process: the drop-in call's own multi-GPU path (devices="all": one
            # host thread and one context per GPU, contiguous row blocks, no process group, no collective) on config 4's
            # rows, 12 500 per GPU, from NumPy arrays.  PCIe-inclusive; beside the RCCL ranks' figure, never `value`.
            # In a child process with a time limit: this path has never run on more than one GPU, and a hang in it
            # must not cost the headline line.
            result["devices_all"] = devices_all_leg(n_visible)

        if world == 1 and not args.no_cpu_baseline:
            # bounded samples of the same workload (for config 5: its X/20000 slice, where the CPU time goes)
            if local_segs is not None:
                p0, p1, b_mode, b_n = local_segs[-1]
                sel = slice(p0, p1)
            else:
                b_mode, b_n, sel = mode, n_points, slice(0, p_gpu)
            d, b, p = den[sel], bmag[sel], bpsi[sel]
            result["cpu_baseline"] = cpu_baseline(freq, alt, d[:512], b[:512], p[:512], b_mode, b_n, what)
            try:
                result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(freq, alt, d[:2048], b[:2048], p[:2048],
                                                                          b_mode, b_n, what)
            except Exception as exc:       # noqa: BLE001 - a sandbox without process spawning must not sink the GPU line
                result["cpu_baseline_all_cores"] = {"error": f"{type(exc).__name__}: {exc}"}
            result["cpu_baseline_fused_c"] = cpu_baseline_c(freq, alt, d, b, p, b_mode, b_n, what)
        os.write(json_fd, (json.dumps(result) + "\n").encode())

    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
