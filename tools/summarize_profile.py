#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles/<tag>.md (+ hbm_traffic.json).

    python tools/summarize_profile.py gpurun_out/prof_r01_v2 r01_v2 [workload-key] [kernel-substring[,substring...]] [description]

Several substrings (a launch made of several kernels - the mixed work list runs a short-grid and a general kernel
side by side): counters are averaged per exact kernel name and then added up; the launch time is then the
HIP-event time bench.py reports (`kernel_ms` in the traced run's JSON line), not a single kernel's duration.

Kernel time comes from `rocprofv3 --kernel-trace --stats`; counters from separate `--pmc`
passes.  HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB
units on this stack and, on gfx950, FETCH_SIZE reports half the bytes of a wide coalesced
read - both the raw and the doubled figure are listed; the doubled one is the bound we quote.
"""

from __future__ import annotations

import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(paths):
    """gpurun merges a new run into an existing local directory: keep, per directory, the newest file only."""
    by_dir = {}
    for path in paths:
        d = os.path.dirname(path)
        if d not in by_dir or os.path.getmtime(path) > os.path.getmtime(by_dir[d]):
            by_dir[d] = path
    return sorted(by_dir.values())


def main():
    src, tag = sys.argv[1], sys.argv[2]
    key = sys.argv[3] if len(sys.argv) > 3 else "X_20000_12500x256"
    kernel = sys.argv[4] if len(sys.argv) > 4 else "vfo_kernel"
    desc = sys.argv[5] if len(sys.argv) > 5 else ("bench.py default workload: 12 500 profiles x 256 freqs, X-mode, "
                                                  "n_points = 20000, fast tier")
    lines = [f"# rocprofv3 summary `{tag}`", "",
             f"Source: `tools/profile.sh {tag}` on one MI355X, workload `{key}` ({desc}); counters are those of the "
             f"dispatches whose kernel name contains `{kernel}`.", ""]

    wanted = [k for k in kernel.split(",") if k]
    selected = lambda name: any(k in name for k in wanted) and "finalize" not in name      # noqa: E731
    stats = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))
    avg_ns = None
    per_kernel_ns = {}
    if stats:
        lines += ["## Kernel trace (`--kernel-trace --stats`)", "",
                  "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(stats[0])):
            lines.append(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | "
                         f"{float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} | {float(r['Percentage']):.3f} |")
            if selected(r["Name"]):
                per_kernel_ns[r["Name"]] = float(r["AverageNs"])
        lines.append("")
    big = {k: v for k, v in per_kernel_ns.items() if v > 0.02 * max(per_kernel_ns.values(), default=0.0)}
    if len(big) == 1:
        avg_ns = next(iter(big.values()))
    elif len(big) > 1:
        # several kernels side by side: the launch time is the HIP-event time of the traced bench run
        log = os.path.join(src, "trace.log")
        if os.path.exists(log):
            for ln in open(log):
                if ln.startswith("{") and '"kernel_ms"' in ln:
                    avg_ns = json.loads(ln)["kernel_ms"] * 1e6
        lines += [f"Launch = {len(big)} kernels side by side ({', '.join('`' + k.split('(')[0] + '`' for k in big)}): rates below use "
                  f"the HIP-event time of the traced run, {avg_ns / 1e6:.3f} ms per launch.", ""] if avg_ns else []
    trace = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")))
    if trace:
        seen = set()
        for r in csv.DictReader(open(trace[0])):
            if selected(r["Kernel_Name"]) and r["Kernel_Name"] not in seen and r["Kernel_Name"] in big:
                seen.add(r["Kernel_Name"])
                lines += [f"`{r['Kernel_Name'].split('(')[0]}`: "] + [f"Dispatch: grid {r['Grid_Size_X']} threads, workgroup {r['Workgroup_Size_X']}, "
                          f"LDS {r['LDS_Block_Size']} B/workgroup, VGPR {r['VGPR_Count']}, "
                          f"SGPR {r['SGPR_Count']}, scratch {r['Scratch_Size']} B.", ""]

    counters = collections.defaultdict(lambda: collections.defaultdict(list))      # counter -> exact kernel name -> values
    for path in newest(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            if selected(r["Kernel_Name"]):
                counters[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # mean per dispatch of each kernel, added up over the kernels of the launch
    mean = {c: sum(sum(v) / len(v) for v in by.values()) for c, by in counters.items()}
    if mean:
        names = sorted({n for by in counters.values() for n in by})
        lines += ["## Counters (separate `--pmc` passes, mean per dispatch" +
                  (", added up over the launch's kernels)" if len(names) > 1 else " of the selected kernel)"), "",
                  "| counter | " + " | ".join("`" + n.split("(")[0].replace("void prhf::", "") + "`" for n in names) +
                  (" | launch |" if len(names) > 1 else " |"), "|---|" + "---|" * (len(names) + (1 if len(names) > 1 else 0))]
        for k in sorted(mean):
            cells = [f"{sum(counters[k][n]) / len(counters[k][n]):.6g}" if n in counters[k] else "-" for n in names]
            lines.append(f"| {k} | " + " | ".join(cells) + (f" | {mean[k]:.6g} |" if len(names) > 1 else " |"))
        lines.append("")

    derived = []
    record = {}
    if avg_ns and "GRBM_GUI_ACTIVE" in mean:
        clk = mean["GRBM_GUI_ACTIVE"] / 8 / (avg_ns * 1e-9) / 1e9
        derived.append(f"* effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time = **{clk:.2f} GHz**")
    if "SQ_ACTIVE_INST_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
        busy = mean["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (mean["GRBM_GUI_ACTIVE"] / 8)
        derived.append(f"* VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / kernel cycles = **{100*busy:.1f} %**")
        record["valu_busy"] = busy
    if "SQ_WAVE_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
        occ = mean["SQ_WAVE_CYCLES"] * 4 / 1024 / (mean["GRBM_GUI_ACTIVE"] / 8)
        derived.append(f"* mean resident waves per SIMD = SQ_WAVE_CYCLES x 4 / 1024 / kernel cycles = **{occ:.2f}**")
    if "SQ_INSTS_VALU" in mean and "SQ_WAVES" in mean:
        derived.append(f"* VALU instructions per wave = {mean['SQ_INSTS_VALU'] / mean['SQ_WAVES']:.4g}; "
                       f"SALU {mean.get('SQ_INSTS_SALU', 0) / mean['SQ_WAVES']:.4g}; "
                       f"LDS {mean.get('SQ_INSTS_LDS', 0) / mean['SQ_WAVES']:.4g}; "
                       f"VMEM reads {mean.get('SQ_INSTS_VMEM_RD', 0) / mean['SQ_WAVES']:.4g}")
    if all(k in mean for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                               "SQ_INSTS_VALU_TRANS_F64")) and avg_ns:
        # per-wave instruction counts x 64 lanes; an FMA is two flops
        flops = 64 * (mean["SQ_INSTS_VALU_ADD_F64"] + mean["SQ_INSTS_VALU_MUL_F64"] +
                      2 * mean["SQ_INSTS_VALU_FMA_F64"] + mean["SQ_INSTS_VALU_TRANS_F64"])
        derived.append(f"* FP64 instructions per dispatch: add {mean['SQ_INSTS_VALU_ADD_F64']:.4g}, mul "
                       f"{mean['SQ_INSTS_VALU_MUL_F64']:.4g}, fma {mean['SQ_INSTS_VALU_FMA_F64']:.4g}, "
                       f"transcendental {mean['SQ_INSTS_VALU_TRANS_F64']:.4g} (wave instructions)")
        derived.append(f"* executed FP64 rate = {flops / (avg_ns * 1e-9) / 1e12:.1f} TFLOP/s "
                       f"(**{100 * flops / (avg_ns * 1e-9) / 78.6e12:.1f} %** of the 78.6 TFLOP/s vector peak; "
                       f"an all-FMA stream would be needed for 100 %)")
        record["fp64_tflops_executed"] = flops / (avg_ns * 1e-9) / 1e12
    if "SQ_LDS_BANK_CONFLICT" in mean and "SQ_LDS_IDX_ACTIVE" in mean:
        derived.append(f"* LDS bank-conflict cycles / LDS active cycles = "
                       f"{100 * mean['SQ_LDS_BANK_CONFLICT'] / mean['SQ_LDS_IDX_ACTIVE']:.2f} %")
    if "TCC_HIT_sum" in mean:
        derived.append(f"* L2 hit rate = {100 * mean['TCC_HIT_sum'] / (mean['TCC_HIT_sum'] + mean['TCC_MISS_sum']):.2f} %")
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        raw = (mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        corrected = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        derived.append(f"* HBM traffic per launch: FETCH_SIZE {mean['FETCH_SIZE']:.0f} KiB + WRITE_SIZE "
                       f"{mean['WRITE_SIZE']:.0f} KiB = {raw/1e6:.1f} MB raw; with the gfx950 x2 read correction "
                       f"**{corrected/1e6:.1f} MB**")
        if avg_ns:
            derived.append(f"* HBM rate = {corrected / (avg_ns * 1e-9) / 1e9:.2f} GB/s "
                           f"({100 * corrected / (avg_ns * 1e-9) / 8e12:.4f} % of 8 TB/s)")
        record.update(hbm_bytes_per_launch=corrected, fetch_kib=mean["FETCH_SIZE"], write_kib=mean["WRITE_SIZE"],
                      kernel_ms=avg_ns / 1e6 if avg_ns else None, source=f"profiles/{tag}.md")
    log = os.path.join(src, "trace.log")
    if os.path.exists(log) and avg_ns:
        for ln in open(log):
            if ln.startswith("{") and '"roofline"' in ln:
                b = json.loads(ln)
                nominal = b["roofline"]["achieved"] * b["kernel_ms"] * 1e-3          # TFLOP per launch (SURVEY 8d count)
                derived.append(f"* nominal roofline (68 flop per reflecting grid point + 6 per level and pair = "
                               f"{nominal:.4g} TFLOP per launch) over this trace's kernel average: "
                               f"**{nominal / (avg_ns * 1e-9):.2f} TFLOP/s = {100 * nominal / (avg_ns * 1e-9) / 78.6:.1f} %** "
                               f"of the 78.6 TFLOP/s FP64 vector peak; {b['value'] * b['ms_per_step'] / (avg_ns / 1e6):.4g} "
                               f"integrals/s at that kernel time")
                record["nominal_tflop_per_launch"] = nominal
    if derived:
        lines += ["## Derived", ""] + derived + [""]

    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", f"{tag}.md"), "w") as fh:
        fh.write("\n".join(lines))
    if record:
        path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = record
        with open(path, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
