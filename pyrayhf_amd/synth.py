"""Seeded synthetic ionospheric profiles (alpha-Chapman F2 + E layers).

These are the inputs of BASELINE.json configs 3-5: PyIRI-shaped
columns on ``alt = arange(80, 700, 1)`` km (N_alt = 620, the shape produced by the
reference's ``generate_input_1D``, reference ``PyRayHF/library.py:2590-2694``), with a
random F2 peak, an E layer that usually leaves an E-F valley, a dipole-like |B| and a
slowly varying field angle.  The draw order below is part of the fixture contract:
changing it changes every seeded batch.
"""

from __future__ import annotations

import numpy as np

N_ALT = 620
ALT_KM = np.arange(80.0, 700.0, 1.0)


def _chapman(nm, hm, scale_h, alt):
    z = (alt[None, :] - hm[:, None]) / scale_h[:, None]
    return nm[:, None] * np.exp(0.5 * (1.0 - z - np.exp(-z)))


def chapman_profiles(n_profiles, seed, rows=None):
    """Return ``(alt (N_alt,), den, bmag, bpsi)`` with the last three ``(P, N_alt)``.

    Units follow the operator: den m^-3, bmag Tesla, bpsi degrees, alt km.  ``rows`` (a
    slice) builds only those rows of the ``n_profiles`` batch - the layer parameters of the
    whole batch are always drawn, so a shard equals the same rows of the full batch.
    """
    rng = np.random.default_rng(seed)
    p = int(n_profiles)
    nmf2 = 10.0 ** rng.uniform(11.3, 12.5, size=p)
    hmf2 = rng.uniform(220.0, 420.0, size=p)
    hf2 = rng.uniform(35.0, 70.0, size=p)
    nme = 10.0 ** rng.uniform(10.3, 11.3, size=p)
    he = rng.uniform(6.0, 12.0, size=p)
    b0 = rng.uniform(2.2e-5, 6.0e-5, size=p)
    psi0 = rng.uniform(0.0, 89.0, size=p)

    if rows is not None:
        nmf2, hmf2, hf2, nme, he, b0, psi0 = (v[rows] for v in (nmf2, hmf2, hf2, nme, he, b0, psi0))
        p = nmf2.size
    alt = ALT_KM.copy()
    den = _chapman(nmf2, hmf2, hf2, alt) + _chapman(nme, np.full(p, 110.0), he, alt)
    bmag = b0[:, None] * ((6371.0 + 80.0) / (6371.0 + alt[None, :])) ** 3
    bpsi = psi0[:, None] + 0.001 * (alt[None, :] - 80.0)
    return alt, np.ascontiguousarray(den), np.ascontiguousarray(bmag), np.ascontiguousarray(bpsi)


def chapman_profiles_torch(n_profiles, seed, device, rows=None):
    """The batch of ``chapman_profiles`` built on ``device`` with torch: ``(alt, den, bmag, bpsi)`` tensors.

    Same seeded layer parameters (drawn on the host), same formulas; the values equal the NumPy batch up
    to the device's ``exp`` / ``pow`` rounding (~1 ulp).  For large benchmark batches, where building
    100 000 profiles with NumPy takes longer than timing them; tests and fixtures use the NumPy version.
    """
    import torch

    rng = np.random.default_rng(seed)
    p = int(n_profiles)
    draws = [10.0 ** rng.uniform(11.3, 12.5, size=p), rng.uniform(220.0, 420.0, size=p), rng.uniform(35.0, 70.0, size=p),
             10.0 ** rng.uniform(10.3, 11.3, size=p), rng.uniform(6.0, 12.0, size=p), rng.uniform(2.2e-5, 6.0e-5, size=p),
             rng.uniform(0.0, 89.0, size=p)]
    if rows is not None:
        draws = [v[rows] for v in draws]
    nmf2, hmf2, hf2, nme, he, b0, psi0 = (torch.as_tensor(v, device=device)[:, None] for v in draws)
    alt = torch.as_tensor(ALT_KM, device=device)

    def chapman(nm, hm, scale_h):
        z = (alt[None, :] - hm) / scale_h
        return nm * torch.exp(0.5 * (1.0 - z - torch.exp(-z)))

    den = chapman(nmf2, hmf2, hf2) + chapman(nme, 110.0, he)
    bmag = b0 * ((6371.0 + 80.0) / (6371.0 + alt[None, :])) ** 3
    bpsi = psi0 + 0.001 * (alt[None, :] - 80.0)
    return alt, den.contiguous(), bmag.contiguous(), bpsi.contiguous()


def sounder_frequencies(config):
    """Frequency sweeps (MHz) named by BASELINE.json configs."""
    if config in (1, 2, 3, "readme"):
        return np.arange(0.1, 17.5, 0.1)            # reference README.md:49-52, F = 174
    if config == 4:
        return np.linspace(0.5, 16.0, 256)
    if config == 5:
        return np.linspace(0.5, 16.0, 512)
    raise ValueError(f"unknown config {config!r}")
