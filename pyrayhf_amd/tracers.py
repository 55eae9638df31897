"""Stratified Snell's-law ray tracing on the GPU.

``trace_ray_cartesian_snells`` keeps the reference's signature and result dictionary
(reference ``PyRayHF/library.py:1096-1268``); ``trace_rays_cartesian_snells`` traces a batch of
(frequency, elevation[, profile]) rays in one launch, one wavefront per ray.
"""

from __future__ import annotations

import numpy as np

from . import _native
from .library import _as_rows

__all__ = ["trace_ray_cartesian_snells", "trace_rays_cartesian_snells"]

_KEYS = ("group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint", "ground_range_km", "x_turn_km",
         "z_turn_km", "n_path")


def trace_rays_cartesian_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, *, profile_index=None,
                                return_paths=False, device=None):
    """Trace ``R`` rays; ``f0_Hz`` and ``elevation_deg`` broadcast to ``(R,)``.

    ``Ne, Babs, bpsi`` are ``(N_alt,)`` or ``(P, N_alt)`` with ``profile_index`` ``(R,)`` choosing the
    column of each ray; ``alt_km`` ``(N_alt,)`` or ``(P, N_alt)``.  Returns a dict of ``(R,)`` arrays:
    the reference's ``group_path_km, group_delay_sec, x_midpoint, z_midpoint, ground_range_km`` plus the
    turning point ``x_turn_km, z_turn_km`` and ``n_path``; NaN for rays that never turn.  With
    ``return_paths`` also ``x`` and ``z``: ``(R, 2 N_alt + 1)`` padded with NaN.
    """
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
    ctx = _native.context(device)
    rc = ctx.snell_cartesian(f.ctypes.data, e.ctypes.data, idx.ctypes.data if idx is not None else None, f.size,
                             d2.ctypes.data, b2.ctypes.data, p2.ctypes.data, a.ctypes.data, n_prof, n_alt,
                             n_alt if a.ndim == 2 else 0, _native.MODE_O if mode == "O" else _native.MODE_X,
                             out.ctypes.data, px.ctypes.data if return_paths else None,
                             pz.ctypes.data if return_paths else None, stride, 0)
    _native.raise_for(rc)
    res = {k: out[:, i].copy() for i, k in enumerate(_KEYS)}
    res["n_path"] = res["n_path"].astype(np.int64)
    if return_paths:
        res["x"], res["z"] = px, pz
    return res


def trace_ray_cartesian_snells(f0_Hz, elevation_deg, alt_km, Ne, Babs, bpsi, mode, *, device=None):
    """One ray; the reference's signature and result dict (library.py:1096-1268):
    ``x, z`` (path arrays), ``group_path_km, group_delay_sec, x_midpoint, z_midpoint, ground_range_km,
    x_apex_km, z_apex_km`` (the apex entries repeat the midpoint, as in the reference).  A ray that
    never turns returns NaN for every entry."""
    r = trace_rays_cartesian_snells(np.float64(f0_Hz), np.float64(elevation_deg), alt_km, Ne, Babs, bpsi, mode,
                                    return_paths=True, device=device)
    n = int(r["n_path"][0])
    if n == 0:
        return {k: np.nan for k in ("x", "z", "group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint",
                                    "ground_range_km", "x_apex_km", "z_apex_km")}
    return {"x": r["x"][0, :n].copy(), "z": r["z"][0, :n].copy(),
            "group_path_km": float(r["group_path_km"][0]), "group_delay_sec": float(r["group_delay_sec"][0]),
            "x_midpoint": float(r["x_midpoint"][0]), "z_midpoint": float(r["z_midpoint"][0]),
            "ground_range_km": float(r["ground_range_km"][0]),
            "x_apex_km": float(r["x_midpoint"][0]), "z_apex_km": float(r["z_midpoint"][0])}
