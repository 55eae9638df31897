"""Inputs and outputs either side of the operator.

* ``from_input_dict`` takes the dictionary the reference's ``generate_input_1D`` returns
  (reference ``PyRayHF/library.py:2674-2688``: keys ``alt, den, bmag, bpsi`` + metadata) or a list of
  them and returns the arrays the operator consumes.
* ``save_batch_npz`` / ``load_batch_npz`` define the batch wire format: one ``.npz`` with
  ``alt (N_alt,) | (P, N_alt)``, ``den, bmag, bpsi (P, N_alt)`` float64 and optional ``freq (F,)``.
  Arrays only: the reference's own persistence is ``pickle`` (``library.py:2442-2455``), which is
  neither safe to load nor needed.
* ``oblique_to_vertical`` converts an oblique ionogram to its vertical equivalent so that it can be
  compared with the operator's output (reference ``library.py:2697-2742``).
"""

from __future__ import annotations

import numpy as np

__all__ = ["from_input_dict", "save_batch_npz", "load_batch_npz", "oblique_to_vertical"]

_KEYS = ("alt", "den", "bmag", "bpsi")


def from_input_dict(data):
    """``(alt, den, bmag, bpsi)`` float64 arrays from one ``generate_input_1D`` dict (1-D arrays) or a
    sequence of them ((P, N_alt) arrays; ``alt`` stays 1-D when every dict has the same grid)."""
    if isinstance(data, dict):
        missing = [k for k in _KEYS if k not in data]
        if missing:
            raise KeyError(f"input dict lacks {missing}")
        alt, den, bmag, bpsi = (np.ascontiguousarray(np.squeeze(np.asarray(data[k])), dtype=np.float64) for k in _KEYS)
        if not (alt.shape == den.shape == bmag.shape == bpsi.shape and alt.ndim == 1):
            raise ValueError("alt, den, bmag and bpsi must be 1-D arrays of one length")
        return alt, den, bmag, bpsi
    rows = [from_input_dict(d) for d in data]
    if not rows:
        raise ValueError("empty sequence of input dicts")
    n = rows[0][0].size
    if any(r[0].size != n for r in rows):
        raise ValueError("profiles of one batch must share the number of levels")
    alts = np.stack([r[0] for r in rows])
    alt = alts[0] if np.all(alts == alts[0]) else alts
    return alt, np.stack([r[1] for r in rows]), np.stack([r[2] for r in rows]), np.stack([r[3] for r in rows])


def save_batch_npz(path, alt, den, bmag, bpsi, freq=None, **metadata):
    """Write the batch wire format (arrays only; metadata values must be numeric or string)."""
    den, bmag, bpsi = (np.atleast_2d(np.asarray(x, dtype=np.float64)) for x in (den, bmag, bpsi))
    alt = np.asarray(alt, dtype=np.float64)
    if not (den.shape == bmag.shape == bpsi.shape) or alt.shape[-1] != den.shape[1]:
        raise ValueError("den, bmag, bpsi must share a (P, N_alt) shape and alt its last axis")
    payload = dict(alt=alt, den=den, bmag=bmag, bpsi=bpsi)
    if freq is not None:
        payload["freq"] = np.asarray(freq, dtype=np.float64)
    for key, value in metadata.items():
        if key in payload:
            raise ValueError(f"metadata key {key!r} collides with an array name")
        payload[f"meta_{key}"] = np.asarray(value)
    np.savez(path, **payload)


def load_batch_npz(path):
    """Read the batch wire format: dict with ``alt, den, bmag, bpsi`` (+ ``freq`` and ``meta`` if present)."""
    with np.load(path, allow_pickle=False) as z:
        out = {k: np.ascontiguousarray(z[k], dtype=np.float64) for k in _KEYS}
        if "freq" in z.files:
            out["freq"] = np.ascontiguousarray(z["freq"], dtype=np.float64)
        meta = {k[5:]: z[k] for k in z.files if k.startswith("meta_")}
    if meta:
        out["meta"] = meta
    if out["den"].ndim != 2 or not (out["den"].shape == out["bmag"].shape == out["bpsi"].shape):
        raise ValueError("den, bmag, bpsi must be (P, N_alt)")
    return out


def oblique_to_vertical(range_km, group_path_km, freq_oblique_mhz, R_E=6371.):
    """Equivalent vertical frequency [MHz] and midpoint virtual height [km] of an oblique sounding over a
    spherical Earth (reference library.py:2697-2742): secant law with the curvature correction."""
    p = np.asarray(group_path_km)
    f_o = np.asarray(freq_oblique_mhz)
    theta = (range_km / 2.0) / R_E                     # half the central angle between the stations
    sag = R_E * (1.0 - np.cos(theta))                  # Earth-curvature correction [km]
    phi = np.arcsin(range_km / p)                      # incidence angle at the midpoint
    return f_o * np.cos(phi), 0.5 * p * np.cos(phi) - sag
