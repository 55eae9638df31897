#!/usr/bin/env python3
"""Benchmark of the vertical-ionogram forward operator on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one device-resident batch of synthetic profiles:
one fused-kernel launch per GPU (plus, for N > 1, the gather of the result rows over RCCL).
Workload (DESIGN.md "Measurement"): the per-GPU shard of BASELINE.json configs[3] - the
configuration its target "virtual-height integrals/s at n_points=20000 X-mode" is quoted on -
12 500 synthetic Chapman profiles x 256 frequencies, X mode, n_points = 20000 per GPU, so
that N = 8 is exactly config 4 (weak scaling).  The single-profile configs[1] (latency-bound:
174 pairs per launch) is timed too and reported under "single_profile".

Rank 0 prints ONE JSON line.  `value` counts every submitted (profile, frequency) pair; the
reflecting fraction is reported beside it.  Inputs are resident in HBM before the timed region.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X public spec, vector FP64 (the bound that binds)
FLOPS_PER_POINT = 68           # CSE'd nominal FP64 operations per grid point
FLOPS_PER_LEVEL = 6            # reflection search, per bottomside level and pair


def algorithmic_bytes(n_prof, n_alt, n_freq, n_points, per_profile_alt=False):
    """Compulsory HBM traffic of one launch."""
    return 8 * (3 * n_prof * n_alt + n_alt * (n_prof if per_profile_alt else 1) + n_freq + n_points
                + n_prof * n_freq)


def algorithmic_flops(vh, den, n_points):
    """68 * n_points per reflecting pair + 6 * K per pair."""
    k = np.argmax(den, axis=1).astype(np.float64)
    reflecting = np.isfinite(vh).sum(axis=1).astype(np.float64)
    return float((FLOPS_PER_POINT * n_points * reflecting).sum() + (FLOPS_PER_LEVEL * k * vh.shape[1]).sum())


def cpu_baseline(freq, alt, den, bmag, bpsi, mode, n_points, budget_s=20.0):
    """The oracle (NumPy restatement of the reference's CPU path) on a bounded sample."""
    from oracle import vfo_numpy as orc
    done, t0 = 0, time.perf_counter()
    orc.virtual_heights(freq, den[0], bmag[0], bpsi[0], alt, mode, min(n_points, 200))   # import/warm-up
    t0 = time.perf_counter()
    while done < den.shape[0]:
        orc.virtual_heights(freq, den[done], bmag[done], bpsi[done], alt, mode, n_points)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done * freq.size / dt, "unit": "integrals/s", "cores": 1, "kind": "port",
            "sample": f"{done} profiles x {freq.size} freqs, {mode}-mode n_points={n_points}, "
                      f"oracle/vfo_numpy.py single process, {dt:.1f} s"}


def cpu_baseline_c(freq, alt, den, bmag, bpsi, mode, n_points):
    """The fused plain-C restatement (oracle/vfo_oracle.c) on every host core: a stronger
    CPU baseline than the reference's unfused NumPy path.  None when the library is not built."""
    from oracle import vfo_c
    if not vfo_c.available():
        return None
    cores = min(vfo_c.threads(), 16)          # the GPU box's CPU share for one GPU
    vfo_c.virtual_heights_batch(freq, den[:cores], bmag[:cores], bpsi[:cores], alt, mode, 200, n_threads=cores)
    n = min(den.shape[0], 4 * cores)
    t0 = time.perf_counter()
    vfo_c.virtual_heights_batch(freq, den[:n], bmag[:n], bpsi[:n], alt, mode, n_points, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n * freq.size / dt, "unit": "integrals/s", "cores": cores, "kind": "port",
            "sample": f"{n} profiles x {freq.size} freqs, {mode}-mode n_points={n_points}, "
                      f"oracle/vfo_oracle.c OpenMP x{cores}, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--profiles", type=int, default=12500, help="profiles per GPU")
    ap.add_argument("--freqs", type=int, default=256)
    ap.add_argument("--n-points", type=int, default=20000)
    ap.add_argument("--mode", default="X", choices=["O", "X"])
    ap.add_argument("--math", default=None, choices=[None, "faithful", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-profile", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from pyrayhf_amd import _native, library, synth
    from pyrayhf_amd import dist as pdist

    rank, world, local_rank = pdist.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists in the product path)")
    # Rehearsal knob for a one-GPU box: PRHF_BENCH_BACKEND=gloo runs every rank on GPU 0 and gathers
    # through host memory.  The driver's runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get("PRHF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    math = {None: None, "faithful": _native.MATH_FAITHFUL, "fast": _native.MATH_FAST}[args.math]
    p_gpu, n_freq, n_points, mode = args.profiles, args.freqs, args.n_points, args.mode
    p_total = p_gpu * world
    freq = np.linspace(0.5, 16.0, n_freq)                       # config 4 sweep
    lo, hi = pdist.shard_bounds(p_total, world, rank)
    # always the rows of config 4's 100 000-profile draw: N = 1 is its first shard, N = 8 the whole of it
    alt, den, bmag, bpsi = synth.chapman_profiles(max(p_total, 100000), 20260004, rows=slice(lo, hi))
    t = {k: torch.as_tensor(v, device=dev) for k, v in
         (("freq", freq), ("alt", alt), ("den", den), ("bmag", bmag), ("bpsi", bpsi))}
    out = torch.empty((p_gpu, n_freq), dtype=torch.float64, device=dev)
    ctx = _native.context(local_rank)

    def step():
        library.vertical_forward_operator(t["freq"], t["den"], t["bmag"], t["bpsi"], t["alt"], mode, n_points,
                                          math=math, sync=False, out=out)
        ms = ctx.last_kernel_ms()            # HIP events on the launch stream, recorded by the library
        if world > 1:
            pdist.gather_rows(out, p_total)
        return ms

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    _native.raise_for(ctx.sync())
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        vh = out.cpu().numpy()
        ms_step = 1e3 * elapsed / args.steps
        k_ms = float(np.mean(kernel_ms))
        abytes = algorithmic_bytes(p_gpu, alt.size, n_freq, n_points)
        aflops = algorithmic_flops(vh, den, n_points)
        traffic = None
        prof_json = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof_json):
            with open(prof_json) as fh:
                rec = json.load(fh)
            key = f"{mode}_{n_points}_{p_gpu}x{n_freq}"
            traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
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
            "config": {"workload": f"BASELINE configs[3] per-GPU shard: {p_gpu} synthetic Chapman profiles x "
                                   f"{n_freq} freqs (0.5-16 MHz) per GPU, {mode}-mode, n_points={n_points}, "
                                   f"seed 20260004; N=8 is config 4 (100000 x 256)",
                       "profiles_per_gpu": p_gpu, "n_freq": n_freq, "n_points": n_points, "mode": mode,
                       "n_alt": int(alt.size), "math": args.math or "default",
                       "parallelism": f"profile shards x{world}, all_gather of vh rows" if world > 1 else "single GPU"},
            "reflecting_fraction": float(np.isfinite(vh).mean()),
            "workgroups_per_cu": ctx.occupancy(alt.size, _native.MATH_FAST if mode == "X" and math is None
                                               else (math or _native.MATH_FAITHFUL)),
            "kernel_ms": k_ms,
            # the FP64 vector ALU (not MFMA, not HBM) binds this path, so the primary
            # roofline object prices the nominal 68 flop/point against the 78.6 TFLOP/s vector peak;
            # the compulsory-bytes view (arithmetic intensity ~1e4 flop/B) rides along as roofline_hbm.
            "roofline": {"bound": "fp64_valu", "achieved": aflops / (k_ms * 1e-3) / 1e12,
                         "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": aflops / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic,
                         "flops_per_point": FLOPS_PER_POINT,
                         "note": "non-MFMA FP64 vector roofline (SURVEY 8d); traffic = measured HBM bytes/launch"},
            "roofline_hbm": {"bound": "hbm", "achieved": abytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": traffic, "algorithmic_bytes": abytes,
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
            # the same batch handed over as host NumPy buffers (pageable): H2D + kernel + D2H
            # first call: the library's staging arena grows to this batch (hipMalloc); second call: steady state
            library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points, math=math)
            t2 = time.perf_counter()
            library.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points, math=math)
            dt2 = time.perf_counter() - t2
            result["host_buffers"] = {"integrals_per_s": p_gpu * n_freq / dt2, "ms_per_call": 1e3 * dt2,
                                      "note": "PCIe-inclusive (pageable host memory in and out); never `value`"}

        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(freq, alt, den[:64], bmag[:64], bpsi[:64], mode, n_points)   # ~20 s
            result["cpu_baseline_fused_c"] = cpu_baseline_c(freq, alt, den, bmag, bpsi, mode, n_points)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
