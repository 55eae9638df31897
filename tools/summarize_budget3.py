#!/usr/bin/env python3
"""gpurun_out/budget3_<tag>/ (tools/inst_budget3.sh) -> profiles/<tag>_config3_budget.md: per variant of
tools/inst_budget3.py the kernel's time and instruction counters (last repetition of each variant), and the split of
config 3's instructions and time into staging, list, per-item work, loop and queue that the differences give.

    python tools/summarize_budget3.py gpurun_out/budget3_r05 r05
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "vfo_short_kernel<256"


def newest(paths):
    by_dir = {}
    for p in paths:
        d = os.path.dirname(p)
        if d not in by_dir or os.path.getmtime(p) > os.path.getmtime(by_dir[d]):
            by_dir[d] = p
    return sorted(by_dir.values())


def main():
    src, tag = sys.argv[1], sys.argv[2]
    variants = [json.loads(l) for l in open(os.path.join(src, "variants.jsonl")) if l.startswith("{")]
    reps = variants[0]["reps"]
    rows = [dict(v) for v in variants]
    # kernel trace: durations of the selected kernel's dispatches in order
    tr = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")))
    if tr:
        d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
             for r in csv.DictReader(open(tr[0])) if KERNEL in r["Kernel_Name"]]
        d = [x[1] for x in sorted(d)]
        if len(d) == reps * len(rows):
            for i, r in enumerate(rows):
                r["trace_ms"] = min(d[i * reps + 1:(i + 1) * reps])
    counters = []
    for path in newest(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        per = {}
        for r in csv.DictReader(open(path)):
            if KERNEL in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for name, vals in per.items():
            vals = [v for _, v in sorted(vals)]
            if len(vals) == reps * len(rows):
                counters.append(name)
                for i, r in enumerate(rows):
                    r[name] = vals[(i + 1) * reps - 1]
    lines = [f"# Instruction budget of `vfo_short_kernel<256>` on config 3 (`{tag}`)", "",
             "Source: `tools/inst_budget3.sh` on one MI355X - `tools/inst_budget3.py` (10 000 profiles x 174 frequencies, O mode; one "
             f"launch per variant, {reps} repetitions, the last one counted) under `rocprofv3 --kernel-trace` and under "
             "`--pmc` passes of their own.  Wave-level instruction counts per launch.", "",
             "| variant | kernel ms | VALU | SALU | LDS | trans f64 | VALU busy | waves/SIMD |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        ms = r.get("trace_ms", r["kernel_ms"])
        busy = occ = float("nan")
        if "SQ_ACTIVE_INST_VALU" in r and "GRBM_GUI_ACTIVE" in r:
            cyc = r["GRBM_GUI_ACTIVE"] / 8.0
            busy = r["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
            occ = r.get("SQ_WAVE_CYCLES", float("nan")) * 4 / 1024 / cyc
        lines.append(f"| {r['variant']} | {ms:.4f} | {r.get('SQ_INSTS_VALU', float('nan')):.4g} | {r.get('SQ_INSTS_SALU', float('nan')):.4g} | "
                     f"{r.get('SQ_INSTS_LDS', float('nan')):.4g} | {r.get('SQ_INSTS_VALU_TRANS_F64', float('nan')):.4g} | {busy:.3f} | {occ:.2f} |")
    lines.append("")
    def get(prefix):
        for r in rows:
            if r["variant"].startswith(prefix):
                return r
        return None
    one, esc, noq = get("one escaping"), get("174 escaping"), get("config 3, no point queued")
    grids = [r for r in rows if "wave-iterations per item" in r["variant"]]
    full = next((r for r in grids if r["n_points"] == 200), None)
    if one and esc and noq and full and len(grids) >= 3 and all("SQ_INSTS_VALU" in r for r in grids + [one, esc, noq]):
        P = 10000.0
        v = lambda r: r["SQ_INSTS_VALU"] / P
        t = lambda r: r.get("trace_ms", r["kernel_ms"])
        # least squares over the grid sizes with the same lane count as config 3: instructions = fixed + slope x wave-iterations
        same = [r for r in grids if r.get("lanes") == full.get("lanes")]
        xs = [float(r["iterations"]) for r in same]
        n = len(xs)

        def fit(ys):
            mx, my = sum(xs) / n, sum(ys) / n
            slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
            return slope, my - slope * mx
        slope, fixed = fit([v(r) for r in same])
        tslope, tfixed = fit([t(r) for r in same])
        it = full["iterations"]
        lines += [f"## Split of config 3's VALU instructions (per profile, wave instructions; {full.get('lanes')} lanes per pair, {it} wave-iterations per item)", "",
                  f"* straight line through the grid sizes {', '.join(str(r['n_points']) for r in same)}: **{slope:.0f}** per profile and wave-iteration "
                  f"(all items of a profile together) + **{fixed:.0f}** that do not depend on the grid",
                  f"* of the fixed part: staging (argmax, nodes, running maximum) + one pass of the list **{v(one):.0f}**; the candidate list "
                  f"over 174 frequencies + **{v(esc) - v(one):.0f}**; per-item set-up, reductions, the queue's bookkeeping and its entries "
                  f"(at least one per pair: the point at the reflection height), the final sums **{fixed - v(esc):.0f}**",
                  f"* config 3: {it} x {slope:.0f} = **{it * slope:.0f}** in the loop of **{v(full):.0f}** in all ({full['SQ_INSTS_VALU']:.4g} per launch); "
                  f"what the ill-conditioned points cost (default minus `well_conditioned = 0`, which queues nothing): **{v(full) - v(noq):.0f}**", "",
                  "## The same split in time (ms per launch; phases overlap across the workgroups of a CU: a difference is what a phase ADDS to the launch)", "",
                  f"* staging + one list pass {t(one):.3f}; list over 174 frequencies {t(esc) - t(one):+.3f}; straight line through the grid sizes: "
                  f"{tfixed:.3f} + {tslope * 1e3:.1f} us per wave-iteration; config 3 {t(full):.3f}.  (The `well_conditioned = 0` launch, "
                  f"{t(noq):.3f} ms, is an instruction count only: with nothing queued the points next to the reflection height stay in the "
                  "reduced algebra, a pair's sum comes out infinite there and its whole item - eight pairs - is evaluated again point by "
                  "point by one wave while the workgroup waits: a serial path that sane settings never take.)", ""]
    out = os.path.join(ROOT, "profiles", f"{tag}_config3_budget.md")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
