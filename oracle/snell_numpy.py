"""CPU oracle for the stratified Snell's-law ray tracers (TEST INFRASTRUCTURE ONLY).

NumPy restatement of the reference's ``trace_ray_cartesian_snells``
(reference ``PyRayHF/library.py:1096-1268``) with its helpers ``tan_from_mu_scalar`` (``:1034-1062``)
and ``find_turning_point`` (``:1065-1093``), built on the oracle's own Appleton-Hartree function.
Same rules as ``oracle/vfo_numpy.py``: only tests, ``smoke()`` and bench baselines import it; it is
pinned bit for bit to vectors that ``oracle/gen_golden.py`` obtained by running the reference
(fixture G8, ``tests/test_oracle_golden.py``).
"""

from __future__ import annotations

import numpy as np

from . import vfo_numpy as vfo

RAY_KEYS = ("x", "z", "group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint", "ground_range_km",
            "x_apex_km", "z_apex_km")


def tangent_from_index(mu_val, p):
    """tan(theta) = p / sqrt(mu^2 - p^2), floored away from the singularity; reference library.py:1034-1062."""
    arg = float(mu_val) ** 2 - p * p
    if arg < 1e-10:
        arg = 1e-10
    return p / np.sqrt(arg)


def turning_altitude(z, mu, p):
    """First bracket where mu falls through the Snell invariant, linear in between; library.py:1065-1093."""
    for i in range(z.size - 1):
        if (mu[i] >= p) and (mu[i + 1] <= p):
            if mu[i] == mu[i + 1]:
                return float(z[i])
            t = (mu[i] - p) / (mu[i] - mu[i + 1])
            return float(z[i] + t * (z[i + 1] - z[i]))
    return np.nan


def with_ground(alt, ne, babs, bpsi):
    """Prepend a level at z = 0 when the grid starts above it; library.py:1173-1182."""
    if alt[0] > 0.0:
        g = [np.interp(0.0, alt, a) for a in (ne, babs, bpsi)]
        return (np.insert(alt, 0, 0.0), np.insert(ne, 0, g[0]), np.insert(babs, 0, g[1]), np.insert(bpsi, 0, g[2]))
    return alt, ne, babs, bpsi


def level_indices(f0_hz, ne, babs, bpsi, mode):
    """mu and mu' on the levels with non-physical values blanked; library.py:1184-1189."""
    with np.errstate(all="ignore"):
        X = vfo.ratio_X(ne, f0_hz)
        Y = vfo.ratio_Y(f0_hz, babs)
        mu, mup = vfo.phase_group_index(X, Y, bpsi, mode)
        mu = np.where((~np.isfinite(mu)) | (mu <= 0.0), np.nan, mu)
        mup = np.where((~np.isfinite(mup)) | (mup <= 0.0), np.nan, mup)
    return mu, mup


def _no_ray():
    return {k: np.nan for k in RAY_KEYS}


def trace_cartesian(f0_hz, elevation_deg, alt_km, ne, babs, bpsi, mode):
    """Flat-Earth stratified Snell's law: up-leg to the turning point, mirrored down-leg, path metrics.

    Reference library.py:1096-1268.  Returns the reference's dict.
    """
    alt, ne, babs, bpsi = with_ground(np.asarray(alt_km, float), np.asarray(ne, float), np.asarray(babs, float),
                                      np.asarray(bpsi, float))
    mu, mup = level_indices(f0_hz, ne, babs, bpsi, mode)
    s0 = np.sin(np.radians(90.0 - elevation_deg))            # :1192-1193
    if not np.isfinite(mu[0]) or not np.isfinite(s0):
        return _no_ray()
    p = mu[0] * s0                                           # :1201
    ok = np.isfinite(mu)
    zv, muv = alt[ok], mu[ok]
    if zv.size < 2:
        return _no_ray()
    z_turn = turning_altitude(zv, muv, p)
    if not np.isfinite(z_turn):
        return _no_ray()
    i_turn = np.searchsorted(zv, z_turn)                     # :1219
    z_up = np.concatenate([zv[:i_turn], [z_turn]])
    mu_up = np.concatenate([muv[:i_turn], [p]])
    x_up = np.zeros_like(z_up)
    if z_up.size > 1:
        dz = np.diff(z_up)
        mu_mid = 0.5 * (mu_up[:-1] + mu_up[1:])
        mu_mid[-1] = max(mu_mid[-1], p + 1e-8)               # :1228
        tan_mid = np.array([tangent_from_index(m, p) for m in mu_mid])
        x_up[1:] = np.cumsum(dz * tan_mid)
    x_turn = x_up[-1]
    x_full = np.concatenate([x_up, ((2.0 * x_turn) - x_up[::-1])[1:]])
    z_full = np.concatenate([z_up, z_up[::-1][1:]])
    with np.errstate(all="ignore"):
        ds = np.hypot(np.diff(x_full), np.diff(z_full))
        path = float(np.nansum(ds))                          # :1242
        mup_path = np.interp(z_full, alt, mup)
        mup_seg = 0.5 * (mup_path[1:] + mup_path[:-1])
        delay = float(np.nansum((mup_seg / vfo.LIGHT_SPEED_KM_S) * ds))   # :1246
    if path > 0:
        mid = int(np.searchsorted(np.cumsum(ds), 0.5 * path))
        x_mid, z_mid = float(x_full[mid]), float(z_full[mid])
    else:
        x_mid = z_mid = np.nan
    landed = float(x_full[-1]) if np.isclose(z_full[-1], 0.0, atol=1e-3) else np.nan
    return {"x": x_full, "z": z_full, "group_path_km": path, "group_delay_sec": delay, "x_midpoint": x_mid,
            "z_midpoint": z_mid, "ground_range_km": landed, "x_apex_km": x_mid, "z_apex_km": z_mid}


def trace_spherical(f0_hz, elevation_deg, alt_km, ne, babs, bpsi, mode="O", *, dz_target_km=1.0, apex_boost=200.0,
                    max_substeps=400, R_E=None):
    """Spherical-Earth stratified Snell's law (Bouguer: mu r sin(theta) = const) with adaptive
    midpoint sub-steps towards the apex.  Reference library.py:1460-1713.  Returns the reference's dict
    (a ray that never turns returns its seven-key NaN dict, library.py:1577-1583)."""
    if R_E is None:
        R_E = vfo.EARTH_RADIUS_KM
    alt, ne, babs, bpsi = with_ground(np.asarray(alt_km, float), np.asarray(ne, float), np.asarray(babs, float),
                                      np.asarray(bpsi, float))
    mu, mup = level_indices(f0_hz, ne, babs, bpsi, mode)
    nothing = {k: np.nan for k in RAY_KEYS[:7]}
    if not np.isfinite(mu[0]):
        return nothing
    p = mu[0] * (R_E + alt[0]) * np.sin(np.radians(90.0 - elevation_deg))      # :1573-1584
    ok = np.isfinite(mu)
    zv, muv = alt[ok], mu[ok]
    if zv.size < 2:
        return nothing
    mu_r = muv * (R_E + zv)
    i0 = next((i for i in range(zv.size - 1) if (mu_r[i] >= p) and (mu_r[i + 1] <= p)), None)   # :1600-1603
    if i0 is None:
        return nothing
    t = (mu_r[i0] - p) / (mu_r[i0] - mu_r[i0 + 1]) if mu_r[i0] != mu_r[i0 + 1] else 0.0
    t = float(np.clip(t, 0.0, 1.0))
    z_turn = zv[i0] + t * (zv[i0 + 1] - zv[i0])                                 # :1619
    z_up = np.concatenate([zv[:i0 + 1], [z_turn]])
    r_up = R_E + z_up
    mu_up = np.concatenate([muv[:i0 + 1], [p / r_up[-1]]])
    phi_up = np.zeros_like(z_up)
    for k in range(len(z_up) - 1):                                              # :1634-1669
        dz = z_up[k + 1] - z_up[k]
        if dz <= 0:
            continue
        mu_a, mu_b = mu_up[k], mu_up[k + 1]
        n_sub = max(1, int(np.ceil(abs(dz) / dz_target_km)))
        gap = min(max(mu_a * r_up[k] - p, 1e-12), max(mu_b * r_up[k + 1] - p, 1e-12))
        n_sub = int(min(max_substeps, n_sub * (1.0 + apex_boost * (1.0 / gap))))
        total = 0.0
        for j in range(n_sub):
            half = 0.5 * (j / n_sub + (j + 1) / n_sub)
            r_m = R_E + (z_up[k] + half * dz)
            mu_r_m = (mu_a + (mu_b - mu_a) * half) * r_m
            if mu_r_m <= p:
                mu_r_m = p + 1e-8
            total += (p / (r_m * np.sqrt(max(mu_r_m * mu_r_m - p * p, 1e-16)))) * (dz / n_sub)
        phi_up[k + 1] = phi_up[k] + total
    phi_full = np.concatenate([phi_up, (2.0 * phi_up[-1] - phi_up[::-1])[1:]])
    z_full = np.concatenate([z_up, z_up[::-1][1:]])
    x_full = R_E * phi_full
    with np.errstate(all="ignore"):
        r_mid = R_E + 0.5 * (z_full[:-1] + z_full[1:])
        ds = np.hypot(r_mid * np.diff(phi_full), np.diff(z_full))               # :1680-1683
        path = float(np.nansum(ds))
        mup_path = np.interp(z_full, alt, mup)
        delay = float(np.nansum((0.5 * (mup_path[:-1] + mup_path[1:]) / vfo.LIGHT_SPEED_KM_S) * ds))
    if path > 0:
        mid = int(np.searchsorted(np.cumsum(ds), 0.5 * path))
        x_mid, z_mid = float(x_full[mid]), float(z_full[mid])
    else:
        x_mid = z_mid = np.nan
    landed = float(x_full[-1]) if np.isclose(z_full[-1], 0.0, atol=1e-3) else np.nan
    return {"x": x_full, "z": z_full, "group_path_km": path, "group_delay_sec": delay, "x_midpoint": x_mid,
            "z_midpoint": z_mid, "ground_range_km": landed, "x_apex_km": x_mid, "z_apex_km": z_mid}
