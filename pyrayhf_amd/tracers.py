"""Stratified Snell's-law ray tracing on the GPU.

``trace_ray_cartesian_snells`` and ``trace_ray_spherical_snells`` keep the reference's signatures
and result dictionaries (reference ``PyRayHF/library.py:1096-1268``, ``:1460-1713``);
``trace_rays_cartesian_snells`` / ``trace_rays_spherical_snells`` trace a batch of
(frequency, elevation[, profile]) rays in one launch, one wavefront per ray;
``trace_fan_cartesian_snells`` / ``trace_fan_spherical_snells`` trace every elevation of a fan for every
frequency (and profile): the refractive-index levels, which depend on the profile and the frequency only, are
computed once per (profile, frequency) and shared by the fan's rays.
"""

from __future__ import annotations

import numpy as np

from . import _native
from .library import MATH_AUTO, _as_rows, constants

__all__ = ["trace_ray_cartesian_snells", "trace_rays_cartesian_snells", "trace_ray_spherical_snells",
           "trace_rays_spherical_snells", "trace_fan_cartesian_snells", "trace_fan_spherical_snells",
           "tan_from_mu_scalar", "find_turning_point"]

_KEYS = ("group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint", "ground_range_km", "x_turn_km",
         "z_turn_km", "n_path")
_DICT_KEYS = ("x", "z", "group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint", "ground_range_km")


def _trace_rays(spherical, f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, profile_index, return_paths, device,
                controls=None, math=None):
    if mode not in ("O", "X"):
        raise ValueError("Mode must be O or X")                     # find_mu_mup, reference library.py:225-226
    f, e = np.broadcast_arrays(np.asarray(f0_Hz, dtype=np.float64), np.asarray(elevation_deg, dtype=np.float64))
    f = np.ascontiguousarray(f).reshape(-1)
    e = np.ascontiguousarray(e).reshape(-1)
    d2, b2, p2 = (np.atleast_2d(_as_rows(n, x)) for n, x in (("Ne", Ne), ("Babs", Babs), ("bpsi", bpsi)))
    if not (d2.shape == b2.shape == p2.shape):
        raise ValueError("Ne, Babs and bpsi must have the same shape")
    n_prof, n_alt = d2.shape
    a = _as_rows("alt_km", alt_km)
    if a.shape[-1] != n_alt or (a.ndim == 2 and a.shape[0] != n_prof):
        raise ValueError("alt_km must have one value per level")
    idx = None
    if profile_index is not None:
        idx = np.ascontiguousarray(np.broadcast_to(np.asarray(profile_index, dtype=np.int64), f.shape))
    elif n_prof != 1:
        raise ValueError("profile_index is needed when several profiles are given")
    out = np.empty((f.size, 8), dtype=np.float64)
    stride = 2 * n_alt + 1
    px = np.empty((f.size, stride), dtype=np.float64) if return_paths else None
    pz = np.empty((f.size, stride), dtype=np.float64) if return_paths else None
    ctx = _native.host_context(device)
    ctx.set_math(MATH_AUTO if math is None else int(math))
    common = (f.ctypes.data, e.ctypes.data, idx.ctypes.data if idx is not None else None, f.size, d2.ctypes.data,
              b2.ctypes.data, p2.ctypes.data, a.ctypes.data, n_prof, n_alt, n_alt if a.ndim == 2 else 0,
              _native.MODE_O if mode == "O" else _native.MODE_X)
    tail = (out.ctypes.data, px.ctypes.data if return_paths else None, pz.ctypes.data if return_paths else None,
            stride, 0)
    rc = ctx.snell_spherical(*common, *controls, *tail) if spherical else ctx.snell_cartesian(*common, *tail)
    _native.raise_for(rc)
    res = {k: out[:, i].copy() for i, k in enumerate(_KEYS)}
    res["n_path"] = res["n_path"].astype(np.int64)
    if return_paths:
        res["x"], res["z"] = px, pz
    return res


def _trace_fan(spherical, f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, return_paths, device, controls, math=None):
    if mode not in ("O", "X"):
        raise ValueError("Mode must be O or X")
    f = np.ascontiguousarray(np.atleast_1d(np.asarray(f0_Hz, dtype=np.float64)))
    e = np.ascontiguousarray(np.atleast_1d(np.asarray(elevation_deg, dtype=np.float64)))
    if f.ndim != 1 or e.ndim != 1:
        raise ValueError("f0_Hz and elevation_deg must be 1-D (frequencies, elevations of the fan)")
    single = np.ndim(Ne) == 1
    d2, b2, p2 = (np.atleast_2d(_as_rows(n, x)) for n, x in (("Ne", Ne), ("Babs", Babs), ("bpsi", bpsi)))
    if not (d2.shape == b2.shape == p2.shape):
        raise ValueError("Ne, Babs and bpsi must have the same shape")
    n_prof, n_alt = d2.shape
    a = _as_rows("alt_km", alt_km)
    if a.shape[-1] != n_alt or (a.ndim == 2 and a.shape[0] != n_prof):
        raise ValueError("alt_km must have one value per level")
    # groups: (profile, frequency) in C order; rays: (profile, frequency, elevation) in C order
    group_f = np.ascontiguousarray(np.tile(f, n_prof))
    group_p = np.ascontiguousarray(np.repeat(np.arange(n_prof, dtype=np.int64), f.size))
    n_groups = group_f.size
    ray_group = np.ascontiguousarray(np.repeat(np.arange(n_groups, dtype=np.int64), e.size))
    ray_e = np.ascontiguousarray(np.tile(e, n_groups))
    n_rays = ray_e.size
    out = np.empty((n_rays, 8), dtype=np.float64)
    stride = 2 * n_alt + 1
    px = np.empty((n_rays, stride), dtype=np.float64) if return_paths else None
    pz = np.empty((n_rays, stride), dtype=np.float64) if return_paths else None
    r_e, dz_t, boost, nsub = controls if spherical else (6371.0, 1.0, 200.0, 400)
    ctx = _native.host_context(device)
    ctx.set_math(MATH_AUTO if math is None else int(math))
    rc = ctx.snell_fan(1 if spherical else 0, group_f.ctypes.data, group_p.ctypes.data, n_groups, ray_group.ctypes.data,
                       ray_e.ctypes.data, n_rays, d2.ctypes.data, b2.ctypes.data, p2.ctypes.data, a.ctypes.data, n_prof,
                       n_alt, n_alt if a.ndim == 2 else 0, _native.MODE_O if mode == "O" else _native.MODE_X,
                       r_e, dz_t, boost, nsub, out.ctypes.data, px.ctypes.data if return_paths else None,
                       pz.ctypes.data if return_paths else None, stride, 0)
    _native.raise_for(rc)
    shape = (f.size, e.size) if single else (n_prof, f.size, e.size)
    res = {k: out[:, i].reshape(shape).copy() for i, k in enumerate(_KEYS)}
    res["n_path"] = res["n_path"].astype(np.int64)
    if return_paths:
        res["x"], res["z"] = px.reshape(shape + (stride,)), pz.reshape(shape + (stride,))
    return res


def trace_fan_cartesian_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, *, return_paths=False, device=None,
                               math=None):
    """Every elevation of ``elevation_deg`` ``(E,)`` for every frequency of ``f0_Hz`` ``(F,)`` (and every profile
    when ``Ne, Babs, bpsi`` are ``(P, N_alt)``), flat Earth.  Returns the dict of ``trace_rays_cartesian_snells``
    with arrays of shape ``(F, E)`` (or ``(P, F, E)``): the values the per-ray call gives for the same rays in the
    reference's operation order (``math=library.MATH_FAITHFUL``), to 1e-13; the level-by-level refractive index is evaluated once per (profile, frequency) instead of once
    per ray (``prhf_snell_fan_f64``)."""
    return _trace_fan(False, f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, return_paths, device, None, math)


def trace_fan_spherical_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode="O", *, dz_target_km=1.0,
                               apex_boost=200.0, max_substeps=400, R_E=None, return_paths=False, device=None, math=None):
    """The same over a spherical Earth, with the reference's apex-refinement controls (library.py:1470-1473)."""
    r_e = constants()[2] if R_E is None else float(R_E)
    return _trace_fan(True, f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, return_paths, device,
                      (r_e, dz_target_km, apex_boost, max_substeps), math)


def _single(r, apex_keys):
    n = int(r["n_path"][0])
    if n == 0:
        return {k: np.nan for k in _DICT_KEYS + (("x_apex_km", "z_apex_km") if apex_keys else ())}
    return {"x": r["x"][0, :n].copy(), "z": r["z"][0, :n].copy(),
            "group_path_km": float(r["group_path_km"][0]), "group_delay_sec": float(r["group_delay_sec"][0]),
            "x_midpoint": float(r["x_midpoint"][0]), "z_midpoint": float(r["z_midpoint"][0]),
            "ground_range_km": float(r["ground_range_km"][0]),
            "x_apex_km": float(r["x_midpoint"][0]), "z_apex_km": float(r["z_midpoint"][0])}


def trace_rays_cartesian_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, *, profile_index=None,
                                return_paths=False, device=None, math=None):
    """Trace ``R`` rays over a flat Earth; ``f0_Hz`` and ``elevation_deg`` broadcast to ``(R,)``.

    ``Ne, Babs, bpsi`` are ``(N_alt,)`` or ``(P, N_alt)`` with ``profile_index`` ``(R,)`` choosing the
    column of each ray; ``alt_km`` ``(N_alt,)`` or ``(P, N_alt)``.  Returns a dict of ``(R,)`` arrays:
    the reference's ``group_path_km, group_delay_sec, x_midpoint, z_midpoint, ground_range_km`` plus the
    turning point ``x_turn_km, z_turn_km`` and ``n_path``; NaN for rays that never turn.  With
    ``return_paths`` also ``x`` and ``z``: ``(R, 2 N_alt + 1)`` padded with NaN.

    ``x_midpoint, z_midpoint``: the reference's search (library.py:1248-1252) lands on the path node before the apex
    or - one rounding away, for one ray in three - on the apex; here always the node before the apex, its
    exact-arithmetic answer (the apex itself is ``x_turn_km, z_turn_km``).

    ``math``: None (default) evaluates the refractive index of a level in the reduced algebra where that cannot move a
    result - far from reflection and from the ray's turning point - and in the reference's operation order elsewhere
    (within 1e-10 of ``library.MATH_FAITHFUL`` - on a spherical Earth one ray in a few million, at the edge of a skip
    zone, up to 3e-10 - which keeps the reference's order at every level and agrees with reference-run rays to 1e-12;
    about three times the time).  Fans always read faithful level tables.
    """
    return _trace_rays(False, f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, profile_index, return_paths,
                       device, math=math)


def trace_rays_spherical_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode="O", *, dz_target_km=1.0,
                                apex_boost=200.0, max_substeps=400, R_E=None, profile_index=None,
                                return_paths=False, device=None, math=None):
    """The same over a spherical Earth (Bouguer's law), with the reference's apex-refinement controls
    (library.py:1470-1473); ``x`` is the ground distance ``R_E * phi``."""
    r_e = constants()[2] if R_E is None else float(R_E)
    return _trace_rays(True, f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, profile_index, return_paths, device,
                       controls=(r_e, dz_target_km, apex_boost, max_substeps), math=math)


def trace_ray_cartesian_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, *, device=None, math=None):
    """One ray; the reference's signature and result dict (library.py:1096-1268):
    ``x, z`` (path arrays), ``group_path_km, group_delay_sec, x_midpoint, z_midpoint, ground_range_km,
    x_apex_km, z_apex_km`` (the apex entries repeat the midpoint, as in the reference).  A ray that
    never turns returns NaN for every entry."""
    r = trace_rays_cartesian_snells(np.float64(f0_Hz), np.float64(elevation_deg), alt_km, Ne, Babs, bpsi, mode,
                                    return_paths=True, device=device, math=math)
    return _single(r, apex_keys=True)


def trace_ray_spherical_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode="O", *, dz_target_km=1.0,
                               apex_boost=200.0, max_substeps=400, R_E=None, device=None, math=None):
    """One ray over a spherical Earth; the reference's signature and result dict (library.py:1460-1713).
    A ray that never turns returns the reference's seven-key NaN dict (library.py:1577-1583)."""
    r = trace_rays_spherical_snells(np.float64(f0_Hz), np.float64(elevation_deg), alt_km, Ne, Babs, bpsi, mode,
                                    dz_target_km=dz_target_km, apex_boost=apex_boost, max_substeps=max_substeps,
                                    R_E=R_E, return_paths=True, device=device, math=math)
    return _single(r, apex_keys=False)


def tan_from_mu_scalar(mu_val, p):
    """Tangent of the ray's angle to the vertical where the phase index is ``mu_val`` and the Snell invariant ``p``
    (reference ``library.py:1034-1062``): ``p / sqrt(max(mu_val**2 - p**2, 1e-10))``.  A host helper beside the
    tracers, which apply the same rule inside the kernel; array arguments broadcast."""
    mu_val = np.asarray(mu_val, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    out = p / np.sqrt(np.maximum(mu_val * mu_val - p * p, 1e-10))
    return float(out) if out.ndim == 0 else out


def find_turning_point(z, mu, p):
    """Altitude at which ``mu`` falls through the Snell invariant ``p``: the first pair of neighbouring nodes with
    ``mu[i] >= p >= mu[i + 1]``, linear in between, NaN when there is none (reference ``library.py:1065-1093``)."""
    z = np.asarray(z, dtype=np.float64).ravel()
    mu = np.asarray(mu, dtype=np.float64).ravel()
    hit = np.nonzero((mu[:-1] >= p) & (mu[1:] <= p))[0]
    if hit.size == 0:
        return float("nan")
    i = int(hit[0])
    if mu[i] == mu[i + 1]:
        return float(z[i])
    return float(z[i] + ((mu[i] - p) / (mu[i] - mu[i + 1])) * (z[i + 1] - z[i]))
