#!/usr/bin/env python3
"""gpurun_out/frow_<tag>/ (tools/f_row_profile.sh) -> profiles/f_row_bounds.json + profiles/<tag>_snell_tracers.md:
per tracer leg of bench.py the vector instructions per ray (SQ_INSTS_VALU of the leg's last dispatch / rays), the
kernel's duration under the trace, VALU-busy share and HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB - the
guide's gfx950 correction).   python tools/f_row_bounds.py gpurun_out/frow_r05 r05"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS = [("snell_per_ray_flat", "snell_kernel", 0), ("snell_per_ray_spherical", "snell_kernel", 1),
        ("snell_fan_flat", "snell_fan_kernel", 0), ("snell_fan_spherical", "snell_fan_kernel", 1)]
REPS = 3


def newest(paths):
    by_dir = {}
    for p in paths:
        d = os.path.dirname(p)
        if d not in by_dir or os.path.getmtime(p) > os.path.getmtime(by_dir[d]):
            by_dir[d] = p
    return sorted(by_dir.values())


def main():
    src, tag = sys.argv[1], sys.argv[2]
    plain = {json.loads(l)["leg"]: json.loads(l) for l in open(os.path.join(src, "plain.jsonl")) if l.startswith("{")}
    counters = {}
    for path in newest(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            counters.setdefault((r["Counter_Name"], r["Kernel_Name"].split("(")[0]), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    durations = {}
    for path in newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))):
        for r in csv.DictReader(open(path)):
            durations.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))

    def pick(table, key_filter, geom):
        """last repetition of geometry `geom` among the dispatches (flat x REPS, spherical x REPS) of the kernels that match"""
        vals = sorted(v for k, lst in table.items() if key_filter(k) for v in lst)
        return vals[(geom + 1) * REPS - 1][1] if len(vals) == 2 * REPS else None

    bounds, lines = {}, [f"# Snell's-law tracer launches of bench.py's f-row legs (`{tag}`)", "",
                         "Source: `tools/f_row_profile.sh` on one MI355X - `tools/f_row_workload.py` (the four launches of bench.py's "
                         "`snell_per_ray` / `snell_fan` legs, three times each) under `rocprofv3 --kernel-trace --stats` and under `--pmc` "
                         "passes of their own; the last repetition of each leg is counted.  Default arithmetic: levels far from "
                         "reflection and from the ray's turning point in the reduced algebra (per-ray launch), faithful level tables (fans).", "",
                         "| leg | rays | kernel ms (trace) | rays/s (un-profiled run) | VALU / ray | SALU / ray | LDS / ray | VALU busy | waves/SIMD | HBM MB per launch |",
                         "|---|---|---|---|---|---|---|---|---|---|"]
    for leg, kernel, geom in LEGS:
        sel = lambda name: kernel in name and ("fan" in kernel or "fan" not in name)      # noqa: E731
        c = {n: pick(counters, lambda k, n=n: k[0] == n and sel(k[1]), geom) for n in
             ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE")}
        ms = pick(durations, sel, geom)
        rays = plain[leg]["rays"]
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if c["FETCH_SIZE"] is not None and c["WRITE_SIZE"] is not None else None
        busy = occ = float("nan")
        if c["GRBM_GUI_ACTIVE"] and c["SQ_ACTIVE_INST_VALU"]:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            busy = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
            occ = c["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc
        if c["SQ_INSTS_VALU"]:
            bounds[leg] = {"valu_per_ray": c["SQ_INSTS_VALU"] / rays, "rays": rays, "kernel_ms_under_trace": ms,
                           "hbm_bytes_per_launch": hbm, "valu_busy": busy,
                           "source": f"profiles/{tag}_snell_tracers.md (rocprofv3 SQ_INSTS_VALU / rays; FETCH_SIZE x 2 + WRITE_SIZE)"}
        lines.append(f"| `{leg}` | {rays} | {ms if ms is None else round(ms, 4)} | {plain[leg]['rays_per_s']:.3e} | "
                     f"{(c['SQ_INSTS_VALU'] or 0) / rays:.0f} | {(c['SQ_INSTS_SALU'] or 0) / rays:.0f} | {(c['SQ_INSTS_LDS'] or 0) / rays:.0f} | "
                     f"{busy:.3f} | {occ:.2f} | {hbm / 1e6 if hbm else float('nan'):.1f} |")
    lines += ["", "The per-ray and fan kernels are issue-bound, not HBM-bound (a launch moves a few hundred MB in about a millisecond): "
              "bench.py prices a leg as `valu_per_ray x rays/s` against 1024 SIMDs x 2.4 GHz / 4 cycles = 6.14e11 wave instructions/s.", ""]
    open(os.path.join(ROOT, "profiles", "f_row_bounds.json"), "w").write(json.dumps(bounds, indent=1) + "\n")
    open(os.path.join(ROOT, "profiles", f"{tag}_snell_tracers.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
