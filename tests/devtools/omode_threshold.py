"""How the default O-mode arithmetic depends on its threshold (context option "well_conditioned": the reduced algebra where
1 - X exceeds it, the reference's operation order below): error distribution against the NumPy oracle and kernel time
per threshold, each in a process of its own.
Usage: python tests/devtools/omode_threshold.py            (driver)
       python tests/devtools/omode_threshold.py worker     (one threshold, from the environment)"""
import sys, os, json, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np
CASES = ((200, 400), (2000, 200), (20000, 40))
CACHE = "/tmp/omode_threshold_oracle.npz"

def inputs():
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003, rows=slice(0, 400))
    return synth.sounder_frequencies(3), alt, den, bmag, bpsi

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    from pyrayhf_amd import library, _native
    freq, alt, den, bmag, bpsi = inputs()
    if os.environ.get("SWEEP_WELL_CONDITIONED"):
        library.set_option("well_conditioned", float(os.environ["SWEEP_WELL_CONDITIONED"]))
    want = np.load(CACHE)
    for n, rows in CASES:
        ms = []
        for _ in range(3):
            got = library.vertical_forward_operator(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n)
            ms.append(library.last_kernel_ms())
        ref = library.vertical_forward_operator(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n, math=library.MATH_FAITHFUL)
        w = want[f"n{n}"]
        ok = np.isfinite(w) & np.isfinite(got)
        err = np.abs(got[ok] - w[ok]) / np.abs(w[ok])
        dref = np.abs(got[ok] - ref[ok]) / np.abs(w[ok])
        print(json.dumps({"threshold": os.environ.get("SWEEP_WELL_CONDITIONED", "default 1e-5"), "n_points": n, "pairs": int(ok.sum()),
                          "mask_diffs": int((np.isnan(got) != np.isnan(w)).sum()), "within_1e-6": float((err <= 1e-6).mean()),
                          "p99": float(np.percentile(err, 99)), "max": float(err.max()),
                          "vs_reference_order": {"p99": float(np.percentile(dref, 99)), "max": float(dref.max()), "over_1e-7": int((dref > 1e-7).sum())},
                          "kernel_ms": min(ms[1:])}), flush=True)
else:
    from oracle import vfo_numpy
    freq, alt, den, bmag, bpsi = inputs()
    np.savez(CACHE, **{f"n{n}": vfo_numpy.virtual_heights_batch(freq, den[:rows], bmag[:rows], bpsi[:rows], alt, "O", n) for n, rows in CASES})
    for wc in (None, "3e-6", "1e-6", "3e-7", "1e-7"):
        env = dict(os.environ)
        if wc: env["SWEEP_WELL_CONDITIONED"] = wc
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, check=False)
