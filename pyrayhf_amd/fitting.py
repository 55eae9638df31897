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
from .library import _mode_code, _multiplier, _as_rows, _is_torch, _device_grid, _grid_flag, MATH_AUTO

__all__ = ["residual_VH_batch", "brute_force_fit", "peak_density_from_trace", "brute_grid", "minimize_parameters",
           "resolve_method", "BoundedPair", "pyiri_edp_builder", "model_VH", "residual_VH"]


def pyiri_edp_builder(F2, F1, E, alt, bottom_type='B_bot'):
    """The electron density profile ``(N_alt,)`` the reference's ``model_VH`` builds with PyIRI (library.py:557-586):
    the F1 layer re-derived from the F2 parameters, then the one-level reconstruction (``'B_bot'``) or the continuous
    IRI builder (``'B0_B1'``); the single time / location of the layer dictionaries is picked out (``EDP[0, :, 0]``).
    The default ``edp_builder`` of ``model_VH`` / ``residual_VH`` / ``minimize_parameters``; needs PyIRI, which neither
    the reference nor this package vendors (ImportError otherwise).  ``F1`` is updated in place, as in the reference."""
    try:
        import PyIRI
        import PyIRI.edp_update
        import PyIRI.sh_library
    except ImportError as exc:          # pragma: no cover - PyIRI is absent from the build and test images
        raise ImportError("PyIRI is not installed: pass edp_builder=callable(F2, F1, E, alt, bottom_type) -> EDP (N_alt,)") from exc
    if bottom_type == 'B_bot':
        derived = PyIRI.edp_update.derive_dependent_F1_parameters(F1['P'], F2['Nm'], F2['hm'], F2['B_bot'], E['hm'])
        rebuild = PyIRI.edp_update.reconstruct_density_from_parameters_1level
    elif bottom_type == 'B0_B1':
        derived = PyIRI.sh_library.derive_dependent_F1_parameters(F1['P'], F2['Nm'], F2['hm'], F2['B0'], F2['B1'], E['hm'])
        rebuild = PyIRI.sh_library.EDP_builder_continuous
    else:
        raise ValueError("bottom_type must be 'B_bot' or 'B0_B1'")
    F1['Nm'], F1['fo'], F1['hm'], F1['B_bot'] = derived
    return np.asarray(rebuild(F2, F1, E, alt))[0, :, 0]


def model_VH(F2, F1, E, f_in, alt, b_mag, b_psi, mode='O', n_points=200, bottom_type='B_bot', *, edp_builder=None,
             device=None, math=None):
    """Modeled virtual-height trace and the electron density profile behind it: the reference's ``model_VH``
    (library.py:512-592), same positional arguments and ``(vh, EDP)`` result; the profile comes from ``edp_builder``
    (default: ``pyiri_edp_builder``, the reference's own PyIRI calls), the trace from the GPU operator."""
    from .library import vertical_forward_operator
    build = pyiri_edp_builder if edp_builder is None else edp_builder
    edp = np.asarray(build(F2, F1, E, alt, bottom_type), dtype=np.float64).ravel()
    return vertical_forward_operator(f_in, edp, b_mag, b_psi, alt, mode, n_points, device=device, math=math), edp


def residual_VH(params, F2_init, F1_init, E_init, f_in, vh_obs, alt, b_mag, b_psi, mode='O', n_points=200,
                bottom_type='B_bot', *, edp_builder=None, device=None, math=None):
    """``vh_obs - vh_model`` for one set of layer parameters: the reference's ``residual_VH`` (library.py:595-669).
    ``params`` maps ``'NmF2'``, ``'hmF2'`` and ``'B_bot'`` (or ``'B0'``, ``'B1'``) to numbers or to objects with a
    ``.value`` (lmfit.Parameters); the layer dictionaries are copied, not mutated; modeled NaNs are replaced by
    ``max(nanmean|vh_model|, 100)`` as in the reference.  One candidate per call - a search evaluates its candidates
    together through ``residual_VH_batch``."""
    from copy import deepcopy

    def number(name):
        v = params[name]
        return float(getattr(v, "value", v))
    F2, F1, E = deepcopy(F2_init), deepcopy(F1_init), deepcopy(E_init)
    F2['Nm'] = np.full_like(F2_init['Nm'], number('NmF2'))
    F2['hm'] = np.full_like(F2_init['Nm'], number('hmF2'))
    if bottom_type == 'B_bot':
        F2['B_bot'] = np.full_like(F2_init['Nm'], number('B_bot'))
    elif bottom_type == 'B0_B1':
        F2['B0'] = np.full_like(F2_init['Nm'], number('B0'))
        F2['B1'] = np.full_like(F2_init['Nm'], number('B1'))
    else:
        raise ValueError("bottom_type must be 'B_bot' or 'B0_B1'")
    build = pyiri_edp_builder if edp_builder is None else edp_builder
    edp = np.asarray(build(F2, F1, E, alt, bottom_type), dtype=np.float64).ravel()
    residual = residual_VH_batch(np.atleast_1d(f_in), np.atleast_1d(vh_obs), edp[None, :], b_mag, b_psi, alt, mode, n_points,
                                 device=device, math=math, return_cost=False)
    return residual.ravel()


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

    A GPU-resident torch ``den`` (the other arguments tensors on the same device, or array-likes that are
    uploaded) is used in place and gives tensors on that device: candidates built on the GPU never visit the host.
    """
    code = _mode_code(mode)
    if _is_torch(den) and den.is_cuda:
        return _torch_residual(freq, vh_obs, den, bmag, bpsi, alt, code, n_points, math, return_cost, return_vh)
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
                                       (_native.FLAG_SHARED_FIELD if shared else 0) | _grid_flag(mult, n_points)))
    out = (residual, cost) if return_cost else (residual,)
    if return_vh:
        out = out + (vh,)
    return out if len(out) > 1 else out[0]


def _torch_residual(freq, vh_obs, den, bmag, bpsi, alt, code, n_points, math, return_cost, return_vh):
    import torch

    dev = den.device

    def prep(x, name):
        if not _is_torch(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float64), device=dev)
        if x.device != dev:
            raise ValueError(f"{name} is on {x.device}, expected {dev}")
        return x.to(torch.float64).contiguous()

    f, obs = prep(freq, "freq").reshape(-1), prep(vh_obs, "vh_obs").reshape(-1)
    if f.shape != obs.shape:
        raise ValueError("freq and vh_obs must be 1-D arrays of one length")
    d2 = prep(den, "den")
    d2 = d2.reshape(1, -1) if d2.dim() == 1 else d2
    n_prof, n_alt = d2.shape
    b2, p2, a = prep(bmag, "bmag"), prep(bpsi, "bpsi"), prep(alt, "alt")
    shared = b2.dim() == 1 and p2.dim() == 1
    if shared:
        if b2.shape != (n_alt,) or p2.shape != (n_alt,):
            raise ValueError("bmag and bpsi must have one value per density level")
    elif b2.shape != d2.shape or p2.shape != d2.shape:
        raise ValueError("bmag and bpsi must be (N_alt,) or have den's shape")
    if a.shape != (n_alt,):
        raise ValueError("alt must be 1-D with one value per density level")
    mult, grid_flag = _device_grid((int(n_points),), dev)
    residual = torch.empty((n_prof, f.numel()), dtype=torch.float64, device=dev)
    cost = torch.empty(n_prof, dtype=torch.float64, device=dev)
    vh = torch.empty((n_prof, f.numel()), dtype=torch.float64, device=dev) if return_vh else None
    ctx = _native.context(dev.index if dev.index is not None else torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_math(MATH_AUTO if math is None else int(math))
    _native.raise_for(ctx.vfo_residual(f.data_ptr(), f.numel(), d2.data_ptr(), b2.data_ptr(), p2.data_ptr(), a.data_ptr(),
                                       n_prof, n_alt, n_alt, 0, mult.data_ptr(), int(n_points), code, obs.data_ptr(),
                                       vh.data_ptr() if return_vh else None, residual.data_ptr(), cost.data_ptr(),
                                       _native.FLAG_DEVICE_PTRS | grid_flag | (_native.FLAG_SHARED_FIELD if shared else 0)))
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


def brute_grid(value, percent_sigma=20.0, step=1.0):
    """The nodes lmfit's brute-force search walks for one parameter of ``minimize_parameters``.

    The reference gives the parameter ``min = value - sigma``, ``max = value + sigma`` with
    ``sigma = value * percent_sigma / 100`` and ``brute_step = step`` (library.py:746-757, :783-792).  lmfit
    turns a parameter with a ``brute_step`` into ``slice(min, max, brute_step)`` and hands the slices to
    ``scipy.optimize.brute``, which expands them with ``np.mgrid`` - i.e. ``np.arange(min, max, step)``: the first
    node is ``min`` and ``max`` itself is excluded.  lmfit is not installed in the build container (and not
    vendored by the reference, pyproject.toml:41), so this end-point convention is taken from its documentation:
    **parity unpinned** for the node set itself; everything evaluated AT the nodes is pinned (fixture G11).
    """
    value = float(np.asarray(value).squeeze())
    sigma = value * (float(percent_sigma) / 100.0)
    return np.arange(value - sigma, value + sigma, float(step))


# ---- the reference's other `method` values (library.py:696-704, :794-798: lmfit.minimize(..., method=method)) ---------
# lmfit is not installed here and not vendored by the reference (pyproject.toml:41), so what it does with a method name
# is restated from its documentation ("parity unpinned" for the optimiser's path; every residual it evaluates is the
# pinned one, fixture G11):
#   * names are matched as Minimizer.minimize matches them - 'leasts*' -> MINPACK Levenberg-Marquardt (leastsq),
#     'least_s*' -> scipy least_squares (trust region reflective, bounds handed over as they are), 'brute',
#     'differential_evolution'; any other name is looked up among the scalar minimisers by prefix of lmfit's key or of
#     SciPy's name, and a name that matches nothing runs Nelder-Mead (scalar_minimize's default) - which is what the
#     reference's docstring example "levenberg-marquardt" ends up as;
#   * a parameter with both bounds is optimised through the MINUIT transform lmfit documents ("Bounds implementation"):
#     internal = arcsin(2 (v - min) / (max - min) - 1), v = min + (sin(internal) + 1) (max - min) / 2;
#   * a scalar minimiser sees the sum of squares of the residual array; leastsq runs with lmfit's defaults
#     (ftol = xtol = gtol = 1e-7, maxfev = 2000 (n + 1)); scalar minimisers with maxiter = 1000 (n + 1).
# MI355X side: every optimiser step that needs several residuals at once is ONE launch - the forward-difference
# Jacobian of Levenberg-Marquardt (MINPACK's fdjac2 steps: h = sqrt(eps) |x|), a whole generation of differential
# evolution - through residual_VH_batch; single evaluations go through the same entry point with one candidate.
_SCALAR_METHODS = {
    'nelder': 'Nelder-Mead', 'powell': 'Powell', 'cg': 'CG', 'bfgs': 'BFGS', 'lbfgsb': 'L-BFGS-B', 'tnc': 'TNC',
    'cobyla': 'COBYLA', 'slsqp': 'SLSQP', 'trust-constr': 'trust-constr',
}
# (lmfit's table also lists newton, dogleg, trust-ncg, trust-exact, trust-krylov: they need a user Jacobian / Hessian,
#  which the reference never passes - lmfit raises for them, and so does resolve_method)
_NEEDS_DERIVATIVES = {'newton': 'Newton-CG', 'dogleg': 'dogleg', 'trust-ncg': 'trust-ncg', 'trust-exact': 'trust-exact',
                      'trust-krylov': 'trust-krylov'}
_NOT_RESTATED = ('basinhopping', 'ampgo', 'shgo', 'dual_annealing', 'emcee')


def resolve_method(method):
    """What ``lmfit.minimize(..., method=method)`` runs for a method name, as ``(family, scipy_name)`` with family one of
    ``'brute'``, ``'leastsq'``, ``'least_squares'``, ``'differential_evolution'``, ``'scalar'``."""
    m = str(method).lower()
    if m.startswith('leasts'):
        return 'leastsq', None
    if m.startswith('least_s'):
        return 'least_squares', None
    if m == 'brute':
        return 'brute', None
    if m in _NOT_RESTATED:
        raise NotImplementedError(f"method={method!r}: lmfit's {m} driver is not restated here")
    if m and 'differential_evolution'.startswith(m):
        return 'differential_evolution', None
    chosen = 'Nelder-Mead'                                 # scalar_minimize's default when nothing matches
    for key, val in {**_SCALAR_METHODS, **_NEEDS_DERIVATIVES}.items():
        if m and (key.startswith(m) or val.lower().startswith(m)):
            chosen = val
    if chosen in _NEEDS_DERIVATIVES.values():
        raise NotImplementedError(f"method={method!r} needs a Jacobian the reference never supplies (lmfit raises too)")
    return 'scalar', chosen


class BoundedPair:
    """The two fitted parameters with lmfit's bounds transform: ``min = value - sigma``, ``max = value + sigma``
    (library.py:746-757, :783-792)."""

    def __init__(self, values, sigmas):
        self.value = np.asarray(values, dtype=np.float64)
        sig = np.abs(np.asarray(sigmas, dtype=np.float64))
        self.lo, self.hi = self.value - sig, self.value + sig
        if not np.all(self.hi > self.lo):
            raise ValueError("percent_sigma leaves no room around an initial value")

    def to_internal(self, v):
        return np.arcsin(2.0 * (np.asarray(v, dtype=np.float64) - self.lo) / (self.hi - self.lo) - 1.0)

    def from_internal(self, x):
        return self.lo + (np.sin(np.asarray(x, dtype=np.float64)) + 1.0) * (self.hi - self.lo) / 2.0


def _local_search(residuals, pair, family, scipy_name):
    """Run one of lmfit's local / population optimisers on ``residuals(nodes (P, 2)) -> (P, F)``; returns the fitted
    pair of external values."""
    from scipy import optimize

    def one(v):
        r = residuals(np.asarray(v, dtype=np.float64).reshape(1, 2))[0]
        if not np.isfinite(r).all():
            # lmfit's nan_policy='raise' (its default): a candidate none of whose frequencies reflects gives an all-NaN
            # residual row (library.py:664-665: np.maximum(nanmean of nothing, 100) is NaN)
            raise ValueError("NaN values detected in the output of the residual function: the optimisers cannot handle "
                             "this (lmfit's nan_policy='raise')")
        return r

    if family == 'least_squares':                          # bounds as they are, no transform (lmfit.least_squares)
        sol = optimize.least_squares(one, pair.value, bounds=(pair.lo, pair.hi))
        return sol.x
    if family == 'differential_evolution':                 # bounds as they are; a generation = one launch
        # (a deliberate departure from lmfit, which runs it on the internal variables with updating='immediate',
        # seed=None and polish=True: see minimize_parameters)
        def cost_columns(x):                               # SciPy hands over (2, S)
            r = residuals(np.ascontiguousarray(x.T))
            return np.nan_to_num((r * r).sum(axis=1), nan=np.inf)
        sol = optimize.differential_evolution(cost_columns, list(zip(pair.lo, pair.hi)), vectorized=True,
                                              updating='deferred', seed=0, polish=False)
        return sol.x
    x0 = pair.to_internal(pair.value)
    # the start sits in the middle of its bounds, i.e. at internal 0 - up to the rounding of value -+ sigma.  MINPACK
    # sizes its first trust region by |x0| (100 |D x0|, or 100 when that is 0): a start of 1e-16 instead of 0 gives a
    # region of 1e-11 and the search ends where it began.  Exactly 0 is what the transform means there.
    x0 = np.where(np.abs(x0) < 1e-9, 0.0, x0)
    if family == 'leastsq':
        eps = np.sqrt(np.finfo(np.float64).eps)            # fdjac2 with epsfcn = 0: h = eps |x| (eps where x == 0)

        def jac(x):
            # ... with a floor: the internal variables are angles in [-pi/2, pi/2], and a start in the middle of its
            # bounds is 0 only up to rounding (value - sigma, value + sigma are rounded): MINPACK's rule then steps by
            # 1e-8 x 1e-16, the column comes out as zeros and Levenberg-Marquardt stops where it started
            h = eps * np.maximum(np.abs(x), 1.0)
            pts = np.vstack([x, x + np.diag(h)])           # f(x) and the n forward steps: one launch
            r = residuals(pair.from_internal(pts))
            return ((r[1:] - r[0]) / h[:, None]).T
        x, _, _, _, _ = optimize.leastsq(lambda x: one(pair.from_internal(x)), x0, Dfun=jac, full_output=1, xtol=1e-7,
                                         ftol=1e-7, gtol=1e-7, maxfev=2000 * (x0.size + 1), col_deriv=False)
        return pair.from_internal(x)

    def penalty(x):
        r = one(pair.from_internal(x))
        return float((r * r).sum())
    sol = optimize.minimize(penalty, x0, method=scipy_name, options={'maxiter': 1000 * (x0.size + 1)})
    return pair.from_internal(sol.x)


def minimize_parameters(F2, F1, E, f_in0, vh_obs0, alt, b_mag, b_psi, method='brute', percent_sigma=20., step=1.,
                        mode='O', n_points=200, bottom_type='B_bot', *, edp_builder=None, device=None, math=None):
    """Fit hmF2 and B_bot (or B0) of the F2 layer to an observed trace: the reference's ``minimize_parameters``
    (library.py:672-825) with its brute-force search as ONE batched launch.

    Positional arguments, defaults and the returned triple ``(vh_result, EDP_result, F2_fit)`` are the
    reference's.  What the reference gets from PyIRI inside ``model_VH`` (library.py:557-586) comes from the
    ``edp_builder(F2, F1, E, alt, bottom_type)``, which returns the electron density profile ``(N_alt,)`` [m^-3] of
    the layer dictionaries it is given: by default ``pyiri_edp_builder`` - the reference's own PyIRI calls, so that
    with PyIRI installed the positional call of the reference works unchanged; any other builder (PyIRI is not
    vendored, and absent from the build image) as a keyword.

    As in the reference: observations are filtered to finite values and sorted (:741-745); NmF2 is fixed from
    the highest observed frequency (+0.01 %, :760-778); hmF2 and B_bot (``bottom_type='B_bot'``) or B0
    (``'B0_B1'``) scan ``brute_grid`` around their initial values; every node's residual is ``residual_VH``
    (:636-669: layer parameters written with ``np.full_like(F2['Nm'], ...)``, modeled NaNs replaced by
    ``max(nanmean|vh|, 100)``); the node with the smallest sum of squares wins (the first one on a tie:
    ``scipy.optimize.brute`` takes ``argmin`` of the grid, first parameter outermost); the final trace is
    evaluated at the UNfiltered input frequencies (:821-824).

    ``method``: the reference forwards the name to ``lmfit.minimize`` (:794-798).  ``'brute'`` (default) is the
    batched grid search above.  Other names run what lmfit runs for them (``resolve_method``): Levenberg-Marquardt
    (``'leastsq'``), ``'least_squares'``, the scalar minimisers (``'nelder'``, ``'powell'``, ``'lbfgsb'``, ``'cobyla'``,
    ...; an unknown name such as the docstring's "levenberg-marquardt" is Nelder-Mead, as in lmfit) and
    ``'differential_evolution'``, with lmfit's bounds transform and defaults restated from its documentation; every
    residual they ask for is evaluated by the fused kernel, several at once where the algorithm allows it.  lmfit is
    absent here, so the optimiser's path is parity-unpinned; the residuals are the pinned ones (G11).
    ``'differential_evolution'`` is NOT lmfit's run of it: here the population lives in the external bounds, is
    updated a generation at a time (one launch each), starts from seed 0 and is not polished; lmfit searches its
    internal (arcsine) variables with immediate updating, an unseeded start and an L-BFGS-B polish of the best
    member - expect the same basin, a coarser minimum and another path.
    """
    from copy import deepcopy

    if (bottom_type == 'B_bot') and (F2.get('B_bot') is None):
        raise ValueError('B_bot is not provided in F, but bottom_type is B_bot')                   # :730-732
    if (bottom_type == 'B0_B1') and ((F2.get('B0') is None) or (F2.get('B1') is None)):
        raise ValueError('B0 and B1 are not provided in F, but bottom_type is B0_B1')              # :734-736
    if bottom_type not in ('B_bot', 'B0_B1'):
        raise ValueError("bottom_type must be 'B_bot' or 'B0_B1'")
    family, scipy_name = resolve_method(method)
    if edp_builder is None:
        edp_builder = pyiri_edp_builder
    f_in0 = np.asarray(f_in0, dtype=np.float64)
    vh_obs0 = np.asarray(vh_obs0, dtype=np.float64)
    alt = np.asarray(alt, dtype=np.float64)
    f_in, vh_obs = _sorted_finite(f_in0, vh_obs0)
    if f_in.size == 0:
        raise ValueError("no finite observation")
    old_hmf2 = float(np.asarray(F2['hm']).squeeze())
    second = 'B_bot' if bottom_type == 'B_bot' else 'B0'
    nmf2_new = peak_density_from_trace(f_in[-1], mode, alt=alt, bmag=b_mag, hmf2=old_hmf2)
    hm_nodes = brute_grid(F2['hm'], percent_sigma, step)
    bb_nodes = brute_grid(F2[second], percent_sigma, step)
    if hm_nodes.size == 0 or bb_nodes.size == 0:
        raise ValueError("empty search grid: percent_sigma too small for this step")

    def layers(hm, bb):
        f2 = deepcopy(F2)
        f2['Nm'] = np.full_like(F2['Nm'], nmf2_new)
        f2['hm'] = np.full_like(F2['Nm'], hm)
        f2[second] = np.full_like(F2['Nm'], bb)
        return f2

    def candidates(nodes):
        den = np.empty((len(nodes), alt.size), dtype=np.float64)
        for k, (hm, bb) in enumerate(nodes):
            den[k] = np.asarray(edp_builder(layers(hm, bb), deepcopy(F1), deepcopy(E), alt, bottom_type),
                                dtype=np.float64).ravel()
        return den

    if family == 'brute':
        nodes = [(hm, bb) for hm in hm_nodes for bb in bb_nodes]
        _, cost = residual_VH_batch(f_in, vh_obs, candidates(nodes), b_mag, b_psi, alt, mode, n_points, device=device,
                                    math=math)
        finite = np.isfinite(cost)
        if not finite.any():
            raise ValueError("no node of the search grid produced a finite cost")
        best = int(np.argmin(np.where(finite, cost, np.inf)))
        F2_fit = layers(*nodes[best])
    else:
        sigma = float(percent_sigma) / 100.0
        start = np.array([old_hmf2, float(np.asarray(F2[second]).squeeze())])
        pair = BoundedPair(start, start * sigma)

        def residuals(nodes):
            return residual_VH_batch(f_in, vh_obs, candidates([tuple(v) for v in np.atleast_2d(nodes)]), b_mag, b_psi,
                                     alt, mode, n_points, device=device, math=math, return_cost=False)
        F2_fit = layers(*_local_search(residuals, pair, family, scipy_name))
    EDP_result = np.asarray(edp_builder(deepcopy(F2_fit), deepcopy(F1), deepcopy(E), alt, bottom_type), dtype=np.float64).ravel()
    from .library import vertical_forward_operator
    vh_result = vertical_forward_operator(f_in0, EDP_result, b_mag, b_psi, alt, mode, n_points, device=device, math=math)
    return vh_result, EDP_result, F2_fit
