"""CPU oracle for the vertical-ionogram forward operator (TEST INFRASTRUCTURE ONLY).

This module is a NumPy restatement of the algorithm of the reference hot path
``PyRayHF.library.vertical_forward_operator`` (reference ``PyRayHF/library.py:459-509``
and the helpers it calls, ``library.py:40-438``).  It exists so that the HIP path can
be checked on the GPU box, where the reference itself is not available.

Rules for this file ("Oracle"):

* Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
  may import it.  The product package ``pyrayhf_amd`` never does.
* Parity pin: ``oracle/gen_golden.py`` ran the *reference itself* in the build
  container and stored its outputs under ``tests/golden/``;
  ``tests/test_oracle_golden.py`` requires this restatement to reproduce those
  vectors bit for bit (same NumPy build), stage by stage.

Every function cites the reference lines whose arithmetic it restates.  The order of
floating-point operations inside each expression is deliberately the reference's,
because the O-mode answer is ill-conditioned near reflection (1 - X ~ 1e-9) and any
re-association moves it by up to 1e-5 relative.
"""

from __future__ import annotations

import numpy as np

# reference library.py:61 and :64 (constants()).
PLASMA_CONST = 8.97866275          # Hz m^1.5 : f_N = PLASMA_CONST * sqrt(n_e)
GYRO_CONST = 2.799249247e10        # Hz / T   : f_H = GYRO_CONST * B
EARTH_RADIUS_KM = 6371.0           # reference library.py:67
LIGHT_SPEED_KM_S = 299_792.458     # reference library.py:70

REFLECTION_BACKOFF_KM = 1e-6       # reference library.py:378 (kwarg is overridden there)
GRID_SHARPNESS = 10.0              # reference library.py:363
UNMAGNETISED_TOL = 1e-12           # reference library.py:163 (y_tol)


def plasma_frequency(density):
    """f_N [Hz] from n_e [m^-3]; reference library.py:75-97 (den2freq)."""
    if np.any(np.asarray(density) < 0):
        raise ValueError("Density must be non-negative")
    return np.sqrt(density) * PLASMA_CONST


def ratio_X(density, f_hz):
    """X = (f_N / f)^2 in the reference's sqrt-then-square form; library.py:120-137."""
    return plasma_frequency(density) ** 2 / f_hz ** 2


def ratio_Y(f_hz, b_tesla):
    """Y = f_H / f; reference library.py:140-158."""
    return GYRO_CONST * b_tesla / f_hz


def stretch_multiplier(n_points, sharpness=GRID_SHARPNESS):
    """Exponentially stretched grid on [0, 1], dense near 1.

    Reference library.py:296-321 called with start=0, end=1 (library.py:361-364).
    """
    u = np.linspace(0.0, 1.0, n_points)
    back = 1.0 - u
    factor = (np.exp(sharpness * back) - 1.0) / (np.exp(sharpness) - 1.0)
    return 1.0 - (0 + (1 - 0) * factor)


def bottomside(den, bmag, bpsi, alt):
    """Keep the levels strictly below the density peak; reference library.py:371-375."""
    k_peak = int(np.argmax(den))
    return den[:k_peak], bmag[:k_peak], bpsi[:k_peak], alt[:k_peak]


def reflection_heights(f_hz, den_b, bmag_b, alt_b, mode):
    """Height where the running maximum of X (O) or X+Y (X) first reaches 1.

    Reference library.py:380-407.  Returns (height_minus_backoff, reflects_mask).
    """
    if mode not in ("O", "X"):
        raise ValueError("mode must be 'O' or 'X'")
    f_col = f_hz[:, None]
    X = ratio_X(den_b[None, :], f_col)
    Y = ratio_Y(f_col, bmag_b[None, :])
    cond = X if mode == "O" else X + Y
    run_max = np.maximum.accumulate(cond, axis=1)
    reflects = run_max[:, -1] >= 1.0
    h = np.array([np.interp(1.0, row, alt_b) for row in run_max])
    return np.where(reflects, h - REFLECTION_BACKOFF_KM, np.nan), reflects


def stretched_columns(f_hz, den, bmag, bpsi, alt, mode, n_points):
    """Per-frequency stretched altitude grid and the profile sampled on it.

    Reference library.py:324-438 (regrid_to_nonuniform_grid).  All outputs are
    (n_freq, n_points).
    """
    mult = stretch_multiplier(n_points)
    den_b, bmag_b, bpsi_b, alt_b = bottomside(den, bmag, bpsi, alt)
    h_refl, _ = reflection_heights(f_hz, den_b, bmag_b, alt_b, mode)
    z = mult[None, :] * (h_refl[:, None] - alt_b[0]) + alt_b[0]
    thickness = np.concatenate(
        (np.diff(z, axis=1), np.full((f_hz.size, 1), REFLECTION_BACKOFF_KM)), axis=1)
    flat = z.reshape(-1)
    return {
        "freq": np.repeat(f_hz[:, None], n_points, axis=1),
        "alt": z,
        "dist": thickness,
        "den": np.interp(flat, alt_b, den_b).reshape(z.shape),
        "bmag": np.interp(flat, alt_b, bmag_b).reshape(z.shape),
        "bpsi": np.interp(flat, alt_b, bpsi_b).reshape(z.shape),
        "crit_height": np.repeat(h_refl[:, None], n_points, axis=1),
    }


def _ulp_nudge(rng, a):
    """a with every element moved by -1, 0 or +1 ulp at random (used only by the rounding-noise model)."""
    step = rng.integers(-1, 2, size=np.shape(a))
    up = np.nextafter(a, np.inf)
    down = np.nextafter(a, -np.inf)
    return np.where(step > 0, up, np.where(step < 0, down, a))


def phase_group_index(X, Y, psi_deg, mode, rounding_rng=None):
    """Appleton-Hartree phase index mu and group index mu'.

    Reference library.py:161-256 (find_mu_mup).  psi in degrees.

    ``rounding_rng`` (a numpy Generator; default None = the reference's arithmetic exactly) is for the
    tolerance model only: the results of the four library calls whose last bit differs between math
    libraries - sin, cos, YT**4, YT**3 - are moved by -1/0/+1 ulp at random, which shows how far the
    reference's own answer depends on them (fixture G12, tests/parity.py).
    """
    X = np.asarray(X, dtype=float)
    Y = np.asarray(Y, dtype=float)
    psi_deg = np.asarray(psi_deg, dtype=float)

    # isotropic plasma, library.py:201-207
    if np.nanmax(np.abs(Y)) < UNMAGNETISED_TOL:
        m2 = 1.0 - X
        mu = np.where(m2 > 0.0, np.sqrt(m2), np.nan)
        mup = np.where(np.isfinite(mu) & (mu > 0.0), 1.0 / mu, np.nan)
        return mu, mup

    s = np.sin(np.deg2rad(psi_deg))
    c = np.cos(np.deg2rad(psi_deg))
    if rounding_rng is not None:
        s, c = _ulp_nudge(rounding_rng, s), _ulp_nudge(rounding_rng, c)
    YT = Y * s                                              # library.py:210
    YL = Y * c                                              # library.py:211
    Xm1 = 1.0 - X                                           # library.py:214
    YT4, YT3 = YT ** 4, YT ** 3
    if rounding_rng is not None:
        YT4, YT3 = _ulp_nudge(rounding_rng, YT4), _ulp_nudge(rounding_rng, YT3)
    alpha = 0.25 * YT4 + YL ** 2 * Xm1 ** 2                 # library.py:217
    beta = np.sqrt(alpha)                                   # library.py:218
    if mode == "O":
        sign = 1.0
    elif mode == "X":
        sign = -1.0
    else:
        raise ValueError("Mode must be O or X")             # library.py:225-226
    D = Xm1 - 0.5 * YT ** 2 + sign * beta                   # library.py:229
    radicand = 1.0 - X * Xm1 / D                            # library.py:232
    radicand[radicand < 0] = np.nan                         # library.py:233
    mu = np.sqrt(radicand)
    mu[mu < 0.0] = 0.0                                      # library.py:237
    mu[mu > 1.0] = np.nan                                   # library.py:238

    dbeta_dX = -YL ** 2 * Xm1 / beta                        # library.py:241
    dD_dX = -1.0 + sign * dbeta_dX                          # library.py:242
    dalpha_dY = YT3 * s + 2.0 * YL * Xm1 ** 2 * c           # library.py:244-245
    dbeta_dY = 0.5 * dalpha_dY / beta                       # library.py:246
    dD_dY = -YT * s + sign * dbeta_dY                       # library.py:247
    dmu_dY = (X * Xm1 * dD_dY) / (2.0 * mu * D ** 2)        # library.py:250
    dmu_dX = (1.0 / (2.0 * mu * D)) * (2.0 * X - 1.0 + X * Xm1 / D * dD_dX)  # :251
    mup = mu - (2.0 * X * dmu_dX + Y * dmu_dY)              # library.py:254
    return mu, mup


def group_path(X, Y, psi_deg, thickness, alt_min, mode, rounding_rng=None):
    """Left-rectangle sum of mu' * dh per frequency row; reference library.py:259-293."""
    _, mup = phase_group_index(X, Y, psi_deg, mode, rounding_rng)
    total = np.nansum(mup * thickness, axis=1)
    total[total == 0] = np.nan
    return total + alt_min


def virtual_heights(freq_mhz, den, bmag, bpsi, alt, mode="O", n_points=200, rounding_rng=None):
    """One profile, all frequencies; reference library.py:459-509."""
    with np.errstate(all="ignore"):
        f_hz = np.atleast_1d(np.asarray(freq_mhz, dtype=float)) * 1e6
        den, bmag, bpsi, alt = (np.asarray(a) for a in (den, bmag, bpsi, alt))
        cols = stretched_columns(f_hz, den, bmag, bpsi, alt, mode, n_points)
        X = ratio_X(cols["den"], cols["freq"])
        Y = ratio_Y(cols["freq"], cols["bmag"])
        return group_path(X, Y, cols["bpsi"], cols["dist"], np.min(alt), mode, rounding_rng)


def rounding_noise(freq_mhz, den, bmag, bpsi, alt, mode="O", n_points=200, runs=24, seed=0):
    """max over ``runs`` of |vh' - vh| / |vh| where vh' is the same evaluation with sin, cos, YT**4, YT**3
    moved by -1/0/+1 ulp at random: the reference algorithm's own response to the last bit of its math
    library (inf where the NaN mask flips).  One profile."""
    base = virtual_heights(freq_mhz, den, bmag, bpsi, alt, mode, n_points)
    rng = np.random.default_rng(seed)
    worst = np.zeros_like(base)
    for _ in range(runs):
        v = virtual_heights(freq_mhz, den, bmag, bpsi, alt, mode, n_points, rounding_rng=rng)
        both = np.isfinite(v) & np.isfinite(base)
        rel = np.where(both, np.abs(v - base) / np.abs(np.where(both, base, 1.0)), 0.0)
        rel = np.where(np.isfinite(v) != np.isfinite(base), np.inf, rel)
        worst = np.maximum(worst, rel)
    return worst


def _ulp_jitter(rng, a):
    a = np.asarray(a, dtype=np.float64)
    out = np.nextafter(a, np.where(rng.integers(0, 2, size=a.shape) == 1, np.inf, -np.inf))
    return np.where(a == 0.0, a, out)              # keep exact zeros (densities must stay >= 0)


def noise_floor(freq_mhz, den, bmag, bpsi, alt, mode="O", n_points=200, runs=12, seed=0):
    """Per-pair noise floor of the algorithm itself for (P, N_alt) inputs without a committed fixture: the
    larger of its response to +-1 ulp on every input (what oracle/gen_golden.py records from the reference
    for G4/G5/G10) and to +-1 ulp in sin / cos / YT**4 / YT**3 (rounding_noise), ``runs`` evaluations each."""
    den, bmag, bpsi = (np.atleast_2d(np.asarray(x, dtype=np.float64)) for x in (den, bmag, bpsi))
    alt = np.asarray(alt, dtype=np.float64)
    freq = np.atleast_1d(np.asarray(freq_mhz, dtype=np.float64))
    rng = np.random.default_rng(seed)
    out = np.zeros((den.shape[0], freq.size))
    with np.errstate(all="ignore"):
        for p in range(den.shape[0]):
            a = alt[p] if alt.ndim == 2 else alt
            base = virtual_heights(freq, den[p], bmag[p], bpsi[p], a, mode, n_points)
            worst = rounding_noise(freq, den[p], bmag[p], bpsi[p], a, mode, n_points, runs=runs, seed=seed + p)
            for _ in range(runs):
                v = virtual_heights(_ulp_jitter(rng, freq), _ulp_jitter(rng, den[p]), _ulp_jitter(rng, bmag[p]),
                                    _ulp_jitter(rng, bpsi[p]), _ulp_jitter(rng, a), mode, n_points)
                both = np.isfinite(v) & np.isfinite(base)
                rel = np.where(both, np.abs(v - base) / np.abs(np.where(both, base, 1.0)), 0.0)
                worst = np.maximum(worst, np.where(np.isfinite(v) != np.isfinite(base), np.inf, rel))
            out[p] = worst
    return out


def virtual_heights_batch(freq_mhz, den, bmag, bpsi, alt, mode="O", n_points=200):
    """(P, N_alt) profiles -> (P, F): a plain loop of single-profile evaluations."""
    den = np.atleast_2d(den)
    bmag = np.atleast_2d(bmag)
    bpsi = np.atleast_2d(bpsi)
    alt = np.asarray(alt)
    out = np.empty((den.shape[0], np.atleast_1d(freq_mhz).size))
    for p in range(den.shape[0]):
        a = alt[p] if alt.ndim == 2 else alt
        out[p] = virtual_heights(freq_mhz, den[p], bmag[p], bpsi[p], a, mode, n_points)
    return out


def stage_capture(freq_mhz, den, bmag, bpsi, alt, mode, n_points):
    """Intermediates of every stage, for unit-testing device functions (fixture G6)."""
    with np.errstate(all="ignore"):
        f_hz = np.atleast_1d(np.asarray(freq_mhz, dtype=float)) * 1e6
        cols = stretched_columns(f_hz, den, bmag, bpsi, alt, mode, n_points)
        X = ratio_X(cols["den"], cols["freq"])
        Y = ratio_Y(cols["freq"], cols["bmag"])
        mu, mup = phase_group_index(X, Y, cols["bpsi"], mode)
        out = dict(cols)
        out.update(X=X, Y=Y, mu=mu, mup=mup,
                   vh=group_path(X, Y, cols["bpsi"], cols["dist"], np.min(alt), mode))
        return out


def residual_rows(vh_obs, vh_model):
    """Residual rows of the fitting driver for a (P, F) batch of modeled traces.

    Restates reference library.py:660-669 (residual_VH) row by row: modeled NaNs are replaced by
    max(nanmean(|vh_model|), 100), then residual = vh_obs - vh_model.  Pinned: fixture G11 holds the
    output of the reference's residual_VH itself (its PyIRI-dependent EDP builder replaced the way the
    reference's own test replaces it) and tests/test_oracle_golden.py::test_residual_rows_g11 requires
    this function to reproduce it bit for bit.
    """
    vh_model = np.array(vh_model, dtype=float, copy=True)
    with np.errstate(all="ignore"):
        for row in vh_model:
            row[np.isnan(row)] = np.maximum(np.nanmean(np.abs(row)), 100)
    return np.asarray(vh_obs, dtype=float)[None, :] - vh_model
