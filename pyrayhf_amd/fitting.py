"""Batched residuals and brute-force fitting on top of the forward operator.

The reference fits (hmF2, B_bot) by letting lmfit's brute-force grid search call
``residual_VH`` once per grid node (reference ``PyRayHF/library.py:595-669``, ``:672-825``): every
node builds an electron-density profile with PyIRI, runs ``vertical_forward_operator`` and returns
``vh_obs - vh_model`` with the modeled NaNs replaced.  Here the whole grid is ONE launch: the
candidate profiles go through the fused kernel as a batch and a second small kernel produces the
residual rows and their sums of squares; everything stays in HBM in between.

The profile builder stays with the caller: PyIRI is a third-party dependency that is not vendored
by the reference (``pyproject.toml:43``), so this module takes the candidate densities as an array.
Parity: fixture G11 (``tests/golden/g11_residual.npz``) holds residual rows produced by the reference's
``residual_VH`` itself - its PyIRI-dependent ``model_VH`` replaced, as the reference's own test does
(``test_core.py:345-353``), by a stand-in that calls the reference's operator on an EDP stored in the
fixture; ``tests/test_gpu_fitting.py`` holds this module to those rows.
"""

from __future__ import annotations

import numpy as np

from . import _native
from .library import _mode_code, _multiplier, _as_rows, MATH_AUTO

__all__ = ["residual_VH_batch", "brute_force_fit", "peak_density_from_trace"]


def _sorted_finite(freq, vh_obs):
    """Keep finite observations, sorted by frequency (reference library.py:741-745)."""
    freq = np.asarray(freq, dtype=np.float64).ravel()
    vh_obs = np.asarray(vh_obs, dtype=np.float64).ravel()
    if freq.shape != vh_obs.shape:
        raise ValueError("freq and vh_obs must have the same length")
    good = np.nonzero(np.isfinite(freq + vh_obs))[0]
    order = np.argsort(freq[good])
    return freq[good][order], vh_obs[good][order]


def residual_VH_batch(freq, vh_obs, den, bmag, bpsi, alt, mode='O', n_points=200, *, device=None, math=None,
                      return_cost=True, return_vh=False):
    """Residual rows ``vh_obs - vh_model`` for a batch of candidate density profiles.

    ``den`` is ``(P, N_alt)`` (one row per candidate); ``bmag, bpsi`` are ``(N_alt,)`` (shared, as
    in the reference's fit) or ``(P, N_alt)``; ``alt`` ``(N_alt,)``.  ``freq`` (MHz) and ``vh_obs``
    (km) are used as given (no filtering or sorting).  Returns ``(residual (P, F), cost (P,))`` with
    ``cost = sum(residual**2, axis=1)``, or only ``residual`` when ``return_cost`` is false; with
    ``return_vh`` the modeled traces ``(P, F)`` (NaN where a frequency escapes) are appended.
    """
    code = _mode_code(mode)
    f = np.ascontiguousarray(np.atleast_1d(freq), dtype=np.float64)
    obs = np.ascontiguousarray(np.atleast_1d(vh_obs), dtype=np.float64)
    if f.shape != obs.shape or f.ndim != 1:
        raise ValueError("freq and vh_obs must be 1-D arrays of one length")
    d2 = np.atleast_2d(_as_rows("den", den))
    n_prof, n_alt = d2.shape
    b2, p2 = (_as_rows(n, x) for n, x in (("bmag", bmag), ("bpsi", bpsi)))
    # a fit's candidates share the field (library.py:589-591): one row each, said so to the library - no (P, N_alt)
    # copies on the host, a third of the bytes to upload
    shared = b2.ndim == 1 and p2.ndim == 1
    if shared:
        if b2.shape != (n_alt,) or p2.shape != (n_alt,):
            raise ValueError("bmag and bpsi must have one value per density level")
    else:
        b2, p2 = (np.ascontiguousarray(np.broadcast_to(np.atleast_2d(x), d2.shape)) for x in (b2, p2))
    a = _as_rows("alt", alt)
    if a.shape != (n_alt,):
        raise ValueError("alt must be 1-D with one value per density level")
    mult = _multiplier(n_points)
    residual = np.empty((n_prof, f.size), dtype=np.float64)
    cost = np.empty(n_prof, dtype=np.float64)
    vh = np.empty((n_prof, f.size), dtype=np.float64) if return_vh else None
    ctx = _native.host_context(device)
    ctx.set_math(MATH_AUTO if math is None else int(math))
    # one call: candidates staged once, modeled traces stay in HBM between the two kernels
    _native.raise_for(ctx.vfo_residual(f.ctypes.data, f.size, d2.ctypes.data, b2.ctypes.data, p2.ctypes.data,
                                       a.ctypes.data, n_prof, n_alt, n_alt, 0, mult.ctypes.data, int(n_points), code,
                                       obs.ctypes.data, vh.ctypes.data if return_vh else None,
                                       residual.ctypes.data, cost.ctypes.data,
                                       _native.FLAG_SHARED_FIELD if shared else 0))
    out = (residual, cost) if return_cost else (residual,)
    if return_vh:
        out = out + (vh,)
    return out if len(out) > 1 else out[0]


def brute_force_fit(freq, vh_obs, den_candidates, bmag, bpsi, alt, mode='O', n_points=200, *, device=None,
                    math=None):
    """Grid search over candidate profiles: the node with the smallest sum of squared residuals.

    The batched equivalent of ``lmfit.minimize(residual_VH, ..., method='brute')`` in the
    reference's ``minimize_parameters`` (library.py:794-798), whose objective for an array residual is
    its sum of squares.  Observations are filtered to finite values and sorted by frequency first
    (library.py:741-745).  Returns ``(best_index, cost (P,), vh_best (F_used,), freq_used)``;
    ``vh_best`` is the modeled trace of the best node as the operator returns it (NaN where a
    frequency escapes, like the final ``model_VH`` call of the reference, library.py:821-824), not the
    NaN-filled one the residual was computed from.  Ties go to the first node, as in a grid scan.
    """
    f, obs = _sorted_finite(freq, vh_obs)
    residual, cost, vh = residual_VH_batch(f, obs, den_candidates, bmag, bpsi, alt, mode, n_points, device=device,
                                           math=math, return_vh=True)
    finite = np.isfinite(cost)
    if not finite.any():
        raise ValueError("no candidate profile produced a finite cost")
    best = int(np.argmin(np.where(finite, cost, np.inf)))
    return best, cost, vh[best], f


def peak_density_from_trace(f_max_mhz, mode='O', *, alt=None, bmag=None, hmf2=None):
    """NmF2 implied by the highest observed frequency, raised by 0.01 % so that the last data point
    still reflects (reference library.py:760-778).  X mode needs the field strength at the F2 peak:
    ``alt``, ``bmag`` [T] and the current ``hmf2`` [km]."""
    from .library import constants, freq2den
    f_max_hz = float(f_max_mhz) * 1e6
    if mode == 'O':
        return freq2den(f_max_hz) * 1.0001
    if mode == 'X':
        if alt is None or bmag is None or hmf2 is None:
            raise ValueError("X mode needs alt, bmag and hmf2")
        g_p = constants()[1]
        f_c = np.asarray(bmag, dtype=np.float64)[int(np.argmin(np.abs(np.asarray(alt, dtype=np.float64) - hmf2)))] * g_p
        return freq2den(np.sqrt(f_max_hz ** 2 - f_max_hz * f_c)) * 1.0001     # from X + Y = 1
    raise ValueError("mode must be 'O' or 'X'")
