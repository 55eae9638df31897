#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Build-container only: it reads /root/reference (absent on the GPU box) and writes
nothing but arrays.  No reference source or bytecode is copied (dont_write_bytecode).

The reference package cannot be imported as-is for two ordinary reasons: its
``__init__`` asks importlib.metadata for an installed distribution
(reference PyRayHF/__init__.py:19) and ``library.py:22,24-25`` import lmfit / PyIRI,
which the hot path (library.py:40-509) never touches.  We register empty stand-in
modules for those names and load ``library.py`` by path.

Fixtures:
  G1  reference test_core.py:225-231 inputs, O and X, n_points=50
  G2  reference test_core.py:260-269 (EDP -> vh known answer, O/200)
  G3  reference test_core.py:139-144 mu/mu' known answers + X mode + unmagnetised + find_vh
  G4  Day/Night example profiles x 174 freqs x {O,X} x {200,2000,20000} + noise floors
  G5  seeded synthetic batch (P=64) x 174 x {O/200, X/2000} + noise floors
  G6  stage captures (3 freqs of G4-Day, n_points=50)
  G7  edge cases
  G8/G9  stratified Snell's-law rays, flat and spherical Earth
  G10 the first 64 profiles of BASELINE config 3 (seed 20260003) x 174 freqs, O/200 + noise floors
  G13 profiles of 2 600 / 3 096 levels (more than the kernel keeps in LDS) and NaN-padded densities
  G14 the first 16 profiles of BASELINE config 4 (seed 20260004) x 256 freqs, X/20000 + noise floors
  G15 the first 8 profiles of each slice of BASELINE config 5 (seed 20260005) x 512 freqs, each with its
      slice's mode and n_points (O/200, X/2000, O/2000, X/20000) + noise floors
  G11 residual_VH (library.py:595-669) rows: the reference function itself, with model_VH replaced by
      a stand-in that builds the EDP without PyIRI and calls the reference's own operator
  G12 (made with the ORACLE, not the reference: it needs a hook inside find_mu_mup) "rounding noise" of
      every O-mode fixture above: the response of the restated algorithm - bit-identical to the
      reference on G1-G11 - to -1/0/+1 ulp in the results of sin, cos, YT**4, YT**3

"noise" = max over NOISE_RUNS reference evaluations, each with every input perturbed by
+-1 ulp at random, of |vh' - vh| / |vh| : the reference's own conditioning, used by the
O-mode parity rule.
"""

from __future__ import annotations

import importlib.util
import io
import logging
import os
import pickle
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
NOISE_RUNS = 24


def load_reference_library():
    sys.dont_write_bytecode = True
    for name in ("lmfit", "PyIRI", "PyIRI.sh_library"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["PyIRI"].sh_library = sys.modules["PyIRI.sh_library"]
    pkg = types.ModuleType("PyRayHF")
    pkg.__path__ = []
    pkg.logger = logging.getLogger("PyRayHF_logger")
    sys.modules["PyRayHF"] = pkg
    spec = importlib.util.spec_from_file_location(
        "PyRayHF.library", os.path.join(REF, "PyRayHF", "library.py"))
    lib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lib)
    return lib


class _ArraysOnly(pickle.Unpickler):
    """The example inputs are pickles of NumPy arrays and scalars; allow nothing else."""
    _OK = {("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "_reconstruct"),
           ("numpy", "ndarray"), ("numpy", "dtype"),
           ("numpy._core.multiarray", "scalar"), ("numpy.core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing {module}.{name}")


def load_example(which):
    path = os.path.join(REF, "docs", "tutorials", f"Example_Input_{which}.p")
    with open(path, "rb") as fh:
        d = _ArraysOnly(io.BytesIO(fh.read())).load()
    return {k: np.asarray(d[k], dtype=np.float64) for k in ("alt", "den", "bmag", "bpsi")}


def ulp_jitter(rng, a):
    a = np.asarray(a, dtype=np.float64)
    direction = np.where(rng.integers(0, 2, size=a.shape) == 1, np.inf, -np.inf)
    out = np.nextafter(a, direction)
    return np.where(a == 0.0, a, out)          # keep exact zeros (den >= 0 must hold)


def noise_floor(lib, freq, den, bmag, bpsi, alt, mode, n_points, seed):
    base = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n_points)
    rng = np.random.default_rng(seed)
    worst = np.zeros_like(base)
    for _ in range(NOISE_RUNS):
        v = lib.vertical_forward_operator(ulp_jitter(rng, freq), ulp_jitter(rng, den),
                                          ulp_jitter(rng, bmag), ulp_jitter(rng, bpsi),
                                          ulp_jitter(rng, alt), mode, n_points)
        both = np.isfinite(v) & np.isfinite(base)
        rel = np.where(both, np.abs(v - base) / np.abs(np.where(both, base, 1.0)), 0.0)
        # a NaN mask that flips under 1-ulp jitter is recorded as infinite noise
        rel = np.where(np.isfinite(v) != np.isfinite(base), np.inf, rel)
        worst = np.maximum(worst, rel)
    return base, worst


def main():
    os.makedirs(OUT, exist_ok=True)
    np.seterr(all="ignore")
    lib = load_reference_library()
    from pyrayhf_amd import synth

    # ---- G1 -----------------------------------------------------------------
    freq = np.array([1.0, 2.0, 10.0])
    alt = np.array([100, 200, 300])
    den = np.array([0, 0.5e12, 1e12])
    bmag = np.array([5e-5, 5e-5, 5e-5])
    bpsi = np.array([60.0, 60.0, 60.0])
    np.savez(os.path.join(OUT, "g1_basic.npz"), freq=freq, alt=alt, den=den, bmag=bmag, bpsi=bpsi,
             n_points=50,
             vh_O=lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode="O", n_points=50),
             vh_X=lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode="X", n_points=50))

    # ---- G2: EDP printed in reference test_core.py:267-269 -------------------
    freq = np.array([3.0, 3.5, 3.7])
    edp = np.array([5.39526842e10, 1.77861786e11, 6.66833260e11])
    np.savez(os.path.join(OUT, "g2_edp_kat.npz"), freq=freq, alt=alt, den=edp, bmag=bmag, bpsi=bpsi,
             n_points=200,
             vh_O=lib.vertical_forward_operator(freq, edp, bmag, bpsi, alt),
             vh_published=np.array([236.22215658, 304.53151596, 334.34853791]))

    # ---- G3 -----------------------------------------------------------------
    aX = np.array([0.02926785, 0.70981059, 0.99672596])
    aY = np.array([0.17123449, 0.16205801, 0.15757213])
    psi = np.array([60.91523271, 61.66028645, 62.02450192])
    mu_o, mup_o = lib.find_mu_mup(aX, aY, psi, "O")
    mu_x, mup_x = lib.find_mu_mup(aX, aY, psi, "X")
    uX = np.array([0.5, 1.0, 1.2])
    mu_u, mup_u = lib.find_mu_mup(uX, np.zeros(3), psi, "O")
    vh_small = lib.find_vh(np.array([[0.5, 0.6]]), np.array([[0.1, 0.2]]), np.array([[45.0, 45.0]]),
                           np.array([[1.0, 1.0]]), 100.0, "O")
    np.savez(os.path.join(OUT, "g3_index_kat.npz"), X=aX, Y=aY, psi=psi,
             mu_O=mu_o, mup_O=mup_o, mu_X=mu_x, mup_X=mup_x,
             mu_published=np.array([0.98626092, 0.56890941, 0.06475905]),
             mup_published=np.array([1.01313137, 1.79819741, 19.76001084]),
             unmag_X=uX, unmag_mu=mu_u, unmag_mup=mup_u, find_vh_small=vh_small,
             grid10=lib.smooth_nonuniform_grid(0, 1, 10, 10.0),
             constants=np.array(lib.constants()))

    # ---- G4 -----------------------------------------------------------------
    freq = np.arange(0.1, 17.5, 0.1)
    g4 = {"freq": freq}
    for which in ("Day", "Night"):
        ex = load_example(which)
        for k, v in ex.items():
            g4[f"{which}_{k}"] = v
        for mode in ("O", "X"):
            for n in (200, 2000, 20000):
                vh, nz = noise_floor(lib, freq, ex["den"], ex["bmag"], ex["bpsi"], ex["alt"],
                                     mode, n, seed=sum(map(ord, which + mode)) * 100000 + n)
                g4[f"{which}_{mode}_{n}_vh"] = vh
                g4[f"{which}_{mode}_{n}_noise"] = nz
                print(which, mode, n, "finite", int(np.isfinite(vh).sum()),
                      "noise max", float(np.nanmax(nz[np.isfinite(nz)])), flush=True)
    np.savez(os.path.join(OUT, "g4_day_night.npz"), **g4)

    # ---- G5 -----------------------------------------------------------------
    alt_s, den_s, bmag_s, bpsi_s = synth.chapman_profiles(64, 20260001)
    g5 = {"freq": freq, "alt": alt_s, "den": den_s, "bmag": bmag_s, "bpsi": bpsi_s, "seed": 20260001}
    for mode, n in (("O", 200), ("X", 2000)):
        vh = np.empty((64, freq.size))
        nz = np.empty_like(vh)
        for p in range(64):
            vh[p], nz[p] = noise_floor(lib, freq, den_s[p], bmag_s[p], bpsi_s[p], alt_s, mode, n,
                                       seed=1000 * p + n)
        g5[f"{mode}_{n}_vh"] = vh
        g5[f"{mode}_{n}_noise"] = nz
        print("G5", mode, n, "reflecting fraction", float(np.isfinite(vh).mean()), flush=True)
    np.savez(os.path.join(OUT, "g5_chapman64.npz"), **g5)

    # ---- G6: stage captures ---------------------------------------------------
    day = load_example("Day")
    f3 = freq[[10, 50, 100]]
    g6 = {"freq": f3}
    for mode in ("O", "X"):
        rg = lib.regrid_to_nonuniform_grid(f3 * 1e6, day["den"], day["bmag"], day["bpsi"], day["alt"],
                                           mode=mode, n_points=50)
        X = lib.find_X(rg["den"], rg["freq"])
        Y = lib.find_Y(rg["freq"], rg["bmag"])
        mu, mup = lib.find_mu_mup(X, Y, rg["bpsi"], mode)
        for k in ("den", "bmag", "bpsi", "dist", "alt", "crit_height"):
            g6[f"{mode}_{k}"] = rg[k]
        g6[f"{mode}_X"], g6[f"{mode}_Y"], g6[f"{mode}_mu"], g6[f"{mode}_mup"] = X, Y, mu, mup
        g6[f"{mode}_vh"] = lib.find_vh(X, Y, rg["bpsi"], rg["dist"], np.min(day["alt"]), mode)
    np.savez(os.path.join(OUT, "g6_stages.npz"), **g6)

    # ---- G7: edge cases ---------------------------------------------------------
    g7 = {}

    def case(name, freq, den, bmag, bpsi, alt, n_points, modes=("O", "X")):
        for key, val in (("freq", freq), ("den", den), ("bmag", bmag), ("bpsi", bpsi), ("alt", alt)):
            g7[f"{name}_{key}"] = np.asarray(val, dtype=np.float64)
        g7[f"{name}_n_points"] = n_points
        for mode in modes:
            g7[f"{name}_vh_{mode}"] = lib.vertical_forward_operator(
                np.asarray(freq, dtype=np.float64), np.asarray(den, dtype=np.float64),
                np.asarray(bmag, dtype=np.float64), np.asarray(bpsi, dtype=np.float64),
                np.asarray(alt, dtype=np.float64), mode, n_points)

    alt5 = np.array([100.0, 150.0, 200.0, 250.0, 300.0, 350.0])
    f_edge = np.array([0.5, 1.0, 2.0, 4.0, 6.0, 8.97866275, 9.5])
    # unmagnetised plasma (b == 0): isotropic branch of the group index
    case("b_zero", f_edge, [1e10, 2e11, 5e11, 8e11, 1e12, 9e11], np.zeros(6), np.full(6, 45.0), alt5, 64)
    # exact hit: a level with X == 1.0 exactly for f = 8.97866275 MHz (den = 1e12 -> f_N = f)
    case("exact_hit", f_edge, [1e10, 2e11, 1e12, 1e12, 1.5e12, 1e12], np.full(6, 4e-5),
         np.full(6, 30.0), alt5, 64)
    # bottom of the profile already above cutoff for the lowest frequencies
    case("bottom_above", np.array([0.5, 1.0, 2.0, 5.0, 9.0]), [5e10, 2e11, 5e11, 8e11, 1e12, 9e11],
         np.full(6, 5e-5), np.full(6, 60.0), alt5, 64)
    # density peak at index 1: a single bottomside level
    case("peak_at_1", np.array([0.5, 1.0, 2.0, 5.0]), [1e11, 1e12, 5e11, 4e11, 3e11, 2e11],
         np.full(6, 5e-5), np.full(6, 60.0), alt5, 16)
    # non-uniform altitude grid + E-F valley (running maximum matters)
    alt_nu = np.array([90.0, 95.0, 101.0, 110.0, 124.0, 140.0, 175.0, 200.0, 260.0, 300.0, 340.0, 420.0])
    den_nu = np.array([1e9, 4e10, 1.3e11, 1.1e11, 0.9e11, 1.0e11, 2.4e11, 4e11, 8e11, 1.1e12, 1.2e12, 9e11])
    case("nonuniform", np.arange(0.5, 11.0, 0.25), den_nu, np.linspace(4.8e-5, 4.0e-5, 12),
         np.linspace(25.0, 27.0, 12), alt_nu, 300)
    # two grid points only
    case("two_points", np.array([2.0, 4.0, 7.0]), [1e10, 2e11, 5e11, 8e11, 1e12, 9e11],
         np.full(6, 5e-5), np.full(6, 60.0), alt5, 2)
    # field along the ray (psi = 0) and across it (psi = 90)
    case("psi_0", np.array([2.0, 4.0, 7.0, 8.5]), [1e10, 2e11, 5e11, 8e11, 1e12, 9e11],
         np.full(6, 5e-5), np.zeros(6), alt5, 200)
    case("psi_90", np.array([2.0, 4.0, 7.0, 8.5]), [1e10, 2e11, 5e11, 8e11, 1e12, 9e11],
         np.full(6, 5e-5), np.full(6, 90.0), alt5, 200)
    # plateau in density below the peak (equal neighbouring levels)
    case("plateau", np.array([2.0, 4.0, 6.3, 6.4, 7.0, 8.5]), [1e10, 5e11, 5e11, 5e11, 1e12, 9e11],
         np.full(6, 5e-5), np.full(6, 60.0), alt5, 200)
    np.savez(os.path.join(OUT, "g7_edges.npz"), **g7)

    # error behaviour, recorded as text for the record only
    for bad in ("Z",):
        try:
            lib.vertical_forward_operator(np.array([1.0]), den, bmag, bpsi, alt, bad, 10)
        except Exception as exc:               # noqa: BLE001
            print("mode", bad, "->", type(exc).__name__, exc)
    try:
        lib.vertical_forward_operator(np.array([1.0]), -np.asarray(den, float) - 1, bmag, bpsi, alt, "O", 10)
    except Exception as exc:                   # noqa: BLE001
        print("negative density ->", type(exc).__name__, exc)
    gen_snell(lib)
    gen_config3(lib)
    gen_residual(lib)
    gen_rounding_noise(lib)
    gen_tall(lib)
    gen_config4(lib)
    gen_config5(lib)
    print("fixtures written to", OUT)


def gen_snell(lib):
    """G8: the reference's stratified Snell's-law tracer (library.py:1096-1268) on two profiles:
    the Gaussian layer of reference test_core.py:727-730 (grid starts at the ground) and the Day
    example profile (grid starts at 80 km: the ground level is inserted).  Ragged path arrays are
    stored concatenated with offsets."""
    day = load_example("Day")
    alt_g = np.linspace(0, 600, 200)
    gauss = {"alt": alt_g, "den": 1e12 * np.exp(-(alt_g - 250) ** 2 / (2 * 60 ** 2)),
             "bmag": np.full_like(alt_g, 4e-5), "bpsi": np.full_like(alt_g, 45.0)}
    g8 = {}
    for name, prof in (("gauss", gauss), ("day", day)):
        for k in ("alt", "den", "bmag", "bpsi"):
            g8[f"{name}_{k}"] = prof[k]
        rays, scal, xs, zs, offs = [], [], [], [], [0]
        for mode_i, mode in enumerate(("O", "X")):
            for f_mhz in (2.0, 3.5, 5.0, 7.0, 9.0, 10.0, 12.5, 16.0):
                for elev in (5.0, 20.0, 45.0, 70.0, 85.0, 89.9, 90.0):
                    r = lib.trace_ray_cartesian_snells(f_mhz * 1e6, elev, prof["alt"], prof["den"], prof["bmag"],
                                                       prof["bpsi"], mode)
                    rays.append((mode_i, f_mhz * 1e6, elev))
                    scal.append([r[k] for k in ("group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint",
                                                "ground_range_km", "x_apex_km", "z_apex_km")])
                    x = np.atleast_1d(np.asarray(r["x"], dtype=float))
                    z = np.atleast_1d(np.asarray(r["z"], dtype=float))
                    if x.size == 1 and np.isnan(x[0]):
                        x = z = np.empty(0)
                    xs.append(x)
                    zs.append(z)
                    offs.append(offs[-1] + x.size)
        g8[f"{name}_rays"] = np.array(rays)
        g8[f"{name}_scalars"] = np.array(scal, dtype=float)
        g8[f"{name}_x"] = np.concatenate(xs)
        g8[f"{name}_z"] = np.concatenate(zs)
        g8[f"{name}_offsets"] = np.array(offs)
        print("G8", name, "rays", len(rays), "traced", int(np.isfinite(np.array(scal)[:, 0]).sum()), flush=True)
    # helper known answers, reference test_core.py:613-635
    g8["tan_cases"] = np.array([[2.0, 1.0], [1.0000001, 1.0], [1e-6, 1e-7]])
    g8["tan_values"] = np.array([lib.tan_from_mu_scalar(m, p) for m, p in g8["tan_cases"]])
    np.savez(os.path.join(OUT, "g8_snell.npz"), **g8)

    # G9: the spherical-Earth tracer (library.py:1460-1713) on the same two profiles
    g9 = {}
    for name, prof in (("gauss", gauss), ("day", day)):
        rays, scal, xs, zs, offs = [], [], [], [], [0]
        for mode_i, mode in enumerate(("O", "X")):
            for f_mhz in (2.0, 5.0, 7.0, 10.0, 12.5):
                for elev in (5.0, 20.0, 45.0, 70.0, 89.9, 90.0):
                    r = lib.trace_ray_spherical_snells(f_mhz * 1e6, elev, prof["alt"], prof["den"], prof["bmag"],
                                                       prof["bpsi"], mode)
                    rays.append((mode_i, f_mhz * 1e6, elev))
                    scal.append([r.get(k, np.nan) for k in ("group_path_km", "group_delay_sec", "x_midpoint",
                                                            "z_midpoint", "ground_range_km")])
                    x = np.atleast_1d(np.asarray(r["x"], dtype=float))
                    z = np.atleast_1d(np.asarray(r["z"], dtype=float))
                    if x.size == 1 and np.isnan(x[0]):
                        x = z = np.empty(0)
                    xs.append(x)
                    zs.append(z)
                    offs.append(offs[-1] + x.size)
        g9[f"{name}_rays"] = np.array(rays)
        g9[f"{name}_scalars"] = np.array(scal, dtype=float)
        g9[f"{name}_x"] = np.concatenate(xs)
        g9[f"{name}_z"] = np.concatenate(zs)
        g9[f"{name}_offsets"] = np.array(offs)
        print("G9", name, "rays", len(rays), "traced", int(np.isfinite(np.array(scal)[:, 0]).sum()), flush=True)
    np.savez(os.path.join(OUT, "g9_snell_spherical.npz"), **g9)


def gen_config3(lib):
    """G10: the first 64 profiles of BASELINE config 3 (10 000 Chapman profiles, seed 20260003) x 174 freqs,
    O mode, n_points = 200 - the configuration the O-mode tolerance is about - with the reference's own
    +-1 ulp response per pair."""
    from pyrayhf_amd import synth
    rows = 64
    alt, den, bmag, bpsi = synth.chapman_profiles(10000, 20260003, rows=slice(0, rows))
    freq = synth.sounder_frequencies(3)
    vh = np.empty((rows, freq.size))
    nz = np.empty_like(vh)
    for p in range(rows):
        vh[p], nz[p] = noise_floor(lib, freq, den[p], bmag[p], bpsi[p], alt, "O", 200, seed=20260003 + 7919 * p)
    np.savez(os.path.join(OUT, "g10_config3_rows.npz"), freq=freq, alt=alt, den=den, bmag=bmag, bpsi=bpsi,
             seed=20260003, n_points=200, O_200_vh=vh, O_200_noise=nz)
    fin = np.isfinite(vh)
    print("G10 reflecting fraction", float(fin.mean()), "noise > 1e-6 at", int((nz[fin] > 1e-6).sum()), "of",
          int(fin.sum()), "pairs; max", float(nz[fin & np.isfinite(nz)].max()), flush=True)


_POOL_LIB = None


def _noise_job(job):
    """One row of a seeded batch through the reference + its NOISE_RUNS jittered re-runs (pool worker: the
    reference module is inherited from the parent through fork)."""
    freq, den, bmag, bpsi, alt, mode, n_points, seed = job
    np.seterr(all="ignore")
    return noise_floor(_POOL_LIB, freq, den, bmag, bpsi, alt, mode, n_points, seed)


def _noise_rows(lib, jobs, workers=8):
    import multiprocessing as mp
    global _POOL_LIB
    _POOL_LIB = lib
    with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
        return pool.map(_noise_job, jobs, chunksize=1)


def gen_config4(lib):
    """G14: the first 16 profiles of BASELINE config 4 (100 000 Chapman profiles, seed 20260004) x 256 freqs,
    X mode, n_points = 20000 - the configuration the metric is quoted on - evaluated by the reference
    (library.py:459-509), with its own +-1 ulp response per pair.  Rows 0-15 belong to rank 0's shard of the
    8-rank cut (dist.shard_bounds) and to the 100 000-row launch alike."""
    from pyrayhf_amd import synth
    rows = 16
    alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(0, rows))
    freq = synth.sounder_frequencies(4)
    out = _noise_rows(lib, [(freq, den[p], bmag[p], bpsi[p], alt, "X", 20000, 20260004 + 7919 * p) for p in range(rows)])
    vh = np.array([o[0] for o in out])
    nz = np.array([o[1] for o in out])
    np.savez(os.path.join(OUT, "g14_config4_rows.npz"), freq=freq, alt=alt, den=den, bmag=bmag, bpsi=bpsi,
             seed=20260004, n_points=20000, rows=np.arange(rows), X_20000_vh=vh, X_20000_noise=nz)
    fin = np.isfinite(vh)
    print("G14 reflecting fraction", float(fin.mean()), "noise max", float(nz[fin & np.isfinite(nz)].max()),
          "mask flips under jitter", int(np.isinf(nz).sum()), flush=True)


CONFIG5_SLICES = ((0, 20000, "O", 200), (20000, 35000, "X", 2000), (35000, 45000, "O", 2000), (45000, 50000, "X", 20000))


def gen_config5(lib):
    """G15: the first 8 profiles of each slice of BASELINE config 5 (50 000 Chapman profiles, seed 20260005,
    512 freqs; slices as SURVEY section 8d names them), each evaluated by the reference with its slice's mode and
    n_points, with noise floors.  The rows are global row numbers of the 50 000-row batch: they sit in rank 0's
    cut of every slice (dist.shard_segments) and in the full work list."""
    from pyrayhf_amd import synth
    per = 8
    freq = synth.sounder_frequencies(5)
    g = {"freq": freq, "seed": 20260005, "slices": np.array([(a, b, "OX".index(m), n) for a, b, m, n in CONFIG5_SLICES])}
    jobs, keys = [], []
    for p0, _p1, mode, n in CONFIG5_SLICES:
        alt, den, bmag, bpsi = synth.chapman_profiles(50000, 20260005, rows=slice(p0, p0 + per))
        g["alt"] = alt
        g[f"{mode}_{n}_rows"] = np.arange(p0, p0 + per)
        g[f"{mode}_{n}_den"], g[f"{mode}_{n}_bmag"], g[f"{mode}_{n}_bpsi"] = den, bmag, bpsi
        for p in range(per):
            jobs.append((freq, den[p], bmag[p], bpsi[p], alt, mode, n, 20260005 + 104729 * (p0 + p) + n))
            keys.append((mode, n, p))
    # longest jobs first, so that the pool drains evenly
    order = sorted(range(len(jobs)), key=lambda i: -jobs[i][6])
    out = _noise_rows(lib, [jobs[i] for i in order])
    res = {keys[i]: o for i, o in zip(order, out)}
    for _p0, _p1, mode, n in CONFIG5_SLICES:
        g[f"{mode}_{n}_vh"] = np.array([res[(mode, n, p)][0] for p in range(per)])
        g[f"{mode}_{n}_noise"] = np.array([res[(mode, n, p)][1] for p in range(per)])
        fin = np.isfinite(g[f"{mode}_{n}_vh"])
        nz = g[f"{mode}_{n}_noise"]
        print("G15", mode, n, "reflecting fraction", float(fin.mean()), "noise > 1e-6 at", int((nz[fin] > 1e-6).sum()),
              "of", int(fin.sum()), "; max finite", float(nz[fin & np.isfinite(nz)].max()), flush=True)
    np.savez(os.path.join(OUT, "g15_config5_rows.npz"), **g)
    add_rounding_noise_g15(lib)


def add_rounding_noise_g15(lib):
    """The O-mode rows of G15 also get the rounding noise of G12 (made with the pinned ORACLE, not the reference: the
    response to -1/0/+1 ulp in sin, cos, YT**4, YT**3), stored beside the reference's input-jitter floors."""
    del lib
    from oracle import vfo_numpy as orc
    path = os.path.join(OUT, "g15_config5_rows.npz")
    g = dict(np.load(path))
    for n in (200, 2000):
        g[f"O_{n}_noise_rounding"] = np.array(
            [orc.rounding_noise(g["freq"], g[f"O_{n}_den"][p], g[f"O_{n}_bmag"][p], g[f"O_{n}_bpsi"][p], g["alt"], "O", n,
                                runs=NOISE_RUNS, seed=int(g[f"O_{n}_rows"][p])) for p in range(g[f"O_{n}_den"].shape[0])])
        fin = np.isfinite(g[f"O_{n}_vh"])
        print("G15 O", n, "rounding noise > 1e-6 at", int((g[f"O_{n}_noise_rounding"][fin] > 1e-6).sum()), "of", int(fin.sum()),
              flush=True)
    np.savez(path, **g)


class _Param:
    """What residual_VH reads from an lmfit.Parameters entry (library.py:645-653): `.value`."""
    def __init__(self, value):
        self.value = value


def gen_residual(lib):
    """G11: residual_VH itself (library.py:595-669).  It needs PyIRI only through model_VH (library.py:650),
    which the reference's own test replaces (test_core.py:345-353: patch("PyRayHF.library.model_VH")).  The
    stand-in here turns the F2 parameters residual_VH has just written (Nm, hm, B_bot; library.py:645-649)
    into an alpha-Chapman EDP plus a fixed E layer and calls the REFERENCE's vertical_forward_operator with it,
    exactly as the last lines of model_VH do (library.py:589-591) - so the fixture pins EDP -> operator ->
    NaN fill (library.py:664-665) -> residual (library.py:668)."""
    seen = []

    def model_vh_stand_in(F2, F1, E, f_in, alt, b_mag, b_psi, mode='O', n_points=200, bottom_type='B_bot'):
        z = (alt - F2['hm'].ravel()[0]) / F2['B_bot'].ravel()[0]
        ze = (alt - E['hm'].ravel()[0]) / E['B_bot'].ravel()[0]
        edp = (F2['Nm'].ravel()[0] * np.exp(0.5 * (1.0 - z - np.exp(-z)))
               + E['Nm'].ravel()[0] * np.exp(0.5 * (1.0 - ze - np.exp(-ze))))
        seen.append(edp)
        vh = lib.vertical_forward_operator(f_in, edp, b_mag, b_psi, alt, mode=mode, n_points=n_points)
        return vh, edp

    lib.model_VH = model_vh_stand_in
    one = lambda v: np.array([[[v]]])                                            # noqa: E731
    g11 = {}
    cases = {
        # candidates around a "true" layer: lower NmF2 -> the top frequencies escape (NaN -> filled)
        "grid": dict(alt=np.arange(80.0, 500.0, 1.0), f_in=np.arange(1.0, 9.8, 0.25), E=(3e10, 110.0, 8.0),
                     truth=(1.2e12, 300.0, 45.0),
                     cand=[(nm, hm, bb) for nm in (0.4e12, 0.9e12, 1.2e12, 1.6e12) for hm in (270.0, 300.0, 335.0)
                           for bb in (38.0, 45.0)]),
        # every modeled height NaN (sounder above foF2 of every candidate): nanmean of nothing -> NaN row
        "all_nan": dict(alt=np.arange(80.0, 500.0, 1.0), f_in=np.array([13.0, 14.0, 15.5]), E=(3e10, 110.0, 8.0),
                        truth=(3.0e12, 300.0, 45.0), cand=[(1.0e12, 300.0, 45.0), (1.2e12, 280.0, 40.0)]),
        # low layer: mean |vh| < 100 km -> NaNs filled with 100 (library.py:664-665)
        "low_layer": dict(alt=np.arange(20.0, 160.0, 0.5), f_in=np.arange(1.0, 8.55, 0.5), E=(1e9, 30.0, 4.0),
                          truth=(9e11, 70.0, 9.0), cand=[(4e11, 66.0, 8.0), (9e11, 70.0, 9.0), (6e11, 75.0, 10.0)]),
    }
    for name, c in cases.items():
        alt, f_in = c["alt"], c["f_in"]
        b_mag = 4.6e-5 * ((6371.0 + 80.0) / (6371.0 + alt)) ** 3
        b_psi = 35.0 + 0.002 * (alt - alt[0])
        E = {"Nm": one(c["E"][0]), "hm": one(c["E"][1]), "B_bot": one(c["E"][2])}
        F1 = {"Nm": one(0.0), "hm": one(200.0), "B_bot": one(30.0)}
        for mode, n_points in (("O", 200), ("X", 2000)):
            F2 = {"Nm": one(c["truth"][0]), "hm": one(c["truth"][1]), "B_bot": one(c["truth"][2])}
            vh_obs, _ = model_vh_stand_in(F2, F1, E, f_in, alt, b_mag, b_psi, mode, n_points)
            del seen[:]
            rows = []
            for nm, hm, bb in c["cand"]:
                params = {"NmF2": _Param(nm), "hmF2": _Param(hm), "B_bot": _Param(bb)}
                rows.append(lib.residual_VH(params, F2, F1, E, f_in, vh_obs, alt, b_mag, b_psi, mode=mode,
                                            n_points=n_points, bottom_type='B_bot'))
            g11[f"{name}_{mode}_residual"] = np.array(rows)
            g11[f"{name}_{mode}_vh_obs"] = vh_obs
            g11[f"{name}_{mode}_n_points"] = n_points
            g11[f"{name}_edp"] = np.array(seen)
            print("G11", name, mode, "rows", len(rows), "NaN residuals", int(np.isnan(np.array(rows)).sum()),
                  "escaping obs", int(np.isnan(vh_obs).sum()), flush=True)
            if mode == "O":
                # the modeled traces behind those rows and their noise floors: the REFERENCE re-run NOISE_RUNS times on
                # every candidate EDP with +-1 ulp on every input (noise_floor above), and - made with the pinned
                # oracle, like G12 - the response to +-1 ulp in sin / cos / YT**4 / YT**3
                from oracle import vfo_numpy as orc
                edps = np.array(seen)
                pairs = [noise_floor(lib, f_in, edp, b_mag, b_psi, alt, "O", n_points, seed=1100 + 13 * k)
                         for k, edp in enumerate(edps)]
                g11[f"{name}_O_vh_model"] = np.array([p[0] for p in pairs])
                g11[f"{name}_O_noise"] = np.array([p[1] for p in pairs])
                g11[f"{name}_O_noise_rounding"] = np.array(
                    [orc.rounding_noise(f_in, edp, b_mag, b_psi, alt, "O", n_points, runs=NOISE_RUNS, seed=k)
                     for k, edp in enumerate(edps)])
                fin = np.isfinite(g11[f"{name}_O_vh_model"])
                print("G11", name, "O noise > 1e-6 at", int((g11[f"{name}_O_noise"][fin] > 1e-6).sum()), "of", int(fin.sum()),
                      "; rounding noise > 1e-6 at", int((g11[f"{name}_O_noise_rounding"][fin] > 1e-6).sum()), flush=True)
        for k, v in (("alt", alt), ("freq", f_in), ("bmag", b_mag), ("bpsi", b_psi)):
            g11[f"{name}_{k}"] = v
    g11["cases"] = np.array(sorted(cases))
    np.savez(os.path.join(OUT, "g11_residual.npz"), **g11)


def gen_rounding_noise(lib):
    """G12.  The input-jitter noise floors of G4/G5/G10 miss one thing: NumPy's pow is not correctly rounded
    (~5 % of YT**4 are one ulp off, systematically over whole argument ranges, so +-1 ulp on the INPUTS does
    not flip it), and where D cancels (library.py:229) that one ulp moves a pair by up to 1e-5.  An
    implementation with a different - even an exactly rounding - math library differs from the reference by
    this much, so the parity rule prices it in: max(input noise, rounding noise)."""
    del lib                                                # made with the pinned oracle
    from oracle import vfo_numpy as orc
    g12 = {}
    g4 = dict(np.load(os.path.join(OUT, "g4_day_night.npz")))
    for which in ("Day", "Night"):
        for n in (200, 2000, 20000):
            g12[f"g4_{which}_O_{n}"] = orc.rounding_noise(g4["freq"], g4[f"{which}_den"], g4[f"{which}_bmag"],
                                                           g4[f"{which}_bpsi"], g4[f"{which}_alt"], "O", n,
                                                           runs=NOISE_RUNS, seed=n)
            print("G12", which, n, float(np.nanmax(g12[f"g4_{which}_O_{n}"])), flush=True)
    for name, fname in (("g5", "g5_chapman64.npz"), ("g10", "g10_config3_rows.npz")):
        g = dict(np.load(os.path.join(OUT, fname)))
        g12[f"{name}_O_200"] = np.array([orc.rounding_noise(g["freq"], g["den"][p], g["bmag"][p], g["bpsi"][p],
                                                             g["alt"], "O", 200, runs=NOISE_RUNS, seed=p)
                                         for p in range(g["den"].shape[0])])
        fin = np.isfinite(g["O_200_vh"])
        print("G12", name, "rounding noise > 1e-6 at", int((g12[f"{name}_O_200"][fin] > 1e-6).sum()), "of",
              int(fin.sum()), "; input noise > 1e-6 at", int((g["O_200_noise"][fin] > 1e-6).sum()), flush=True)
    np.savez(os.path.join(OUT, "g12_rounding_noise.npz"), **g12)


def gen_tall(lib):
    """G13.  Profiles of more levels than the kernel keeps in LDS (the reference has no limit, library.py:371-375)
    and densities padded with NaN (np.argmax returns the first NaN, :371: the padding acts as the peak).
      tall_day   the Day example resampled to 0.2 km (3 096 levels), 174 freqs: O/200 with its noise floor, X/2000
      tall_fine  the Day example at 0.1 km (6 191 levels, peak at level 2 580), 34 freqs: O/200, X/2000
      tall_rag   a Chapman layer on 2 600 levels with irregular spacing (0.05 - 0.45 km), 96 freqs: O/200, X/500
      nanpad     the Day example with den = NaN from level 300 up (above the peak at 258) and from level 200 up
                 (below it: the layer is cut short), 174 freqs, O/200 and X/200
      nanfield   the Day example with a NaN in alt, bmag or bpsi at one or two levels (cases in `nanfield_cases`;
                 the inputs are nanpad_* with that column blanked at `nanfield_<case>_levels`), {O, X} x {200, 2000}"""
    from pyrayhf_amd import synth
    g = {}
    d = load_example("Day")
    freq = np.arange(0.1, 17.5, 0.1)
    alt = np.arange(d["alt"][0], d["alt"][-1] + 1e-9, 0.2)
    tall = {k: np.interp(alt, d["alt"], d[k]) for k in ("den", "bmag", "bpsi")}
    g.update(tall_day_freq=freq, tall_day_alt=alt, tall_day_den=tall["den"], tall_day_bmag=tall["bmag"],
             tall_day_bpsi=tall["bpsi"])
    g["tall_day_O_200_vh"], g["tall_day_O_200_noise"] = noise_floor(lib, freq, tall["den"], tall["bmag"], tall["bpsi"],
                                                                    alt, "O", 200, seed=1301)
    g["tall_day_X_2000_vh"] = lib.vertical_forward_operator(freq, tall["den"], tall["bmag"], tall["bpsi"], alt, "X", 2000)
    # ... and at 0.1 km (6 191 levels): its peak sits at level 2 580 - not even the bottomside fits LDS (1 400 levels)
    alt_f = np.arange(d["alt"][0], d["alt"][-1] + 1e-9, 0.1)
    fine = {k: np.interp(alt_f, d["alt"], d[k]) for k in ("den", "bmag", "bpsi")}
    freq_f = np.arange(0.5, 17.5, 0.5)
    g.update(tall_fine_freq=freq_f, tall_fine_alt=alt_f, tall_fine_den=fine["den"], tall_fine_bmag=fine["bmag"],
             tall_fine_bpsi=fine["bpsi"])
    g["tall_fine_O_200_vh"], g["tall_fine_O_200_noise"] = noise_floor(lib, freq_f, fine["den"], fine["bmag"], fine["bpsi"],
                                                                      alt_f, "O", 200, seed=1305)
    g["tall_fine_X_2000_vh"] = lib.vertical_forward_operator(freq_f, fine["den"], fine["bmag"], fine["bpsi"], alt_f, "X", 2000)
    rng = np.random.default_rng(1302)
    alt_r = 80.0 + np.concatenate(([0.0], np.cumsum(rng.uniform(0.05, 0.45, size=2599))))
    a1, den1, bmag1, bpsi1 = synth.chapman_profiles(4, 1302, rows=slice(2, 3))
    rag = {"den": np.interp(alt_r, a1, den1[0]), "bmag": np.interp(alt_r, a1, bmag1[0]),
           "bpsi": np.interp(alt_r, a1, bpsi1[0])}
    freq_r = np.linspace(0.6, 14.0, 96)
    g.update(tall_rag_freq=freq_r, tall_rag_alt=alt_r, tall_rag_den=rag["den"], tall_rag_bmag=rag["bmag"],
             tall_rag_bpsi=rag["bpsi"])
    g["tall_rag_O_200_vh"], g["tall_rag_O_200_noise"] = noise_floor(lib, freq_r, rag["den"], rag["bmag"], rag["bpsi"],
                                                                    alt_r, "O", 200, seed=1303)
    g["tall_rag_X_500_vh"] = lib.vertical_forward_operator(freq_r, rag["den"], rag["bmag"], rag["bpsi"], alt_r, "X", 500)
    g.update(nanpad_freq=freq, nanpad_alt=d["alt"], nanpad_bmag=d["bmag"], nanpad_bpsi=d["bpsi"])
    for first in (300, 200):
        den = d["den"].copy()
        den[first:] = np.nan
        g[f"nanpad_{first}_den"] = den
        g[f"nanpad_{first}_O_200_vh"], g[f"nanpad_{first}_O_200_noise"] = noise_floor(
            lib, freq, den, d["bmag"], d["bpsi"], d["alt"], "O", 200, seed=1304 + first)
        g[f"nanpad_{first}_X_200_vh"] = lib.vertical_forward_operator(freq, den, d["bmag"], d["bpsi"], d["alt"], "X", 200)
    # NaN in the other columns (the operator reproduces the reference's behaviour): the Day example
    # with one or two levels blanked; every case x {O, X} x {200, 2000}
    nan_cases = {"alt_500": ("alt", [500]), "alt_100": ("alt", [100]), "bmag_100": ("bmag", [100]), "bmag_1": ("bmag", [1]),
                 "bpsi_100": ("bpsi", [100]), "bpsi_1": ("bpsi", [1]), "bpsi_50_200": ("bpsi", [50, 200]),
                 "bpsi_257": ("bpsi", [257]), "bmag_400": ("bmag", [400])}
    g["nanfield_cases"] = np.array(sorted(nan_cases))
    for name, (col, levels) in nan_cases.items():
        a = {k: d[k].copy() for k in ("den", "bmag", "bpsi", "alt")}
        a[col][levels] = np.nan
        g[f"nanfield_{name}_col"] = np.array(col)
        g[f"nanfield_{name}_levels"] = np.array(levels)
        for mode in "OX":
            for n in (200, 2000):
                g[f"nanfield_{name}_{mode}_{n}_vh"] = lib.vertical_forward_operator(freq, a["den"], a["bmag"], a["bpsi"],
                                                                                    a["alt"], mode, n)
    np.savez(os.path.join(OUT, "g13_tall_nanpad.npz"), **g)
    for k in sorted(g):
        if k.endswith("_vh"):
            print("G13", k, "finite", int(np.isfinite(g[k]).sum()), "of", g[k].size, flush=True)


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        np.seterr(all="ignore")
        ref = load_reference_library()
        for what in only:
            {"g8": gen_snell, "g10": gen_config3, "g11": gen_residual, "g12": gen_rounding_noise, "g13": gen_tall,
             "g14": gen_config4, "g15": gen_config5, "g15r": add_rounding_noise_g15}[what](ref)
    else:
        main()
