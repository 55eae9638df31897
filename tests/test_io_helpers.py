"""Host-side adapters either side of the operator."""

import numpy as np
import pytest

from conftest import load_golden
from pyrayhf_amd import io as pio
from pyrayhf_amd import library


def test_oblique_to_vertical_identities():
    # reference test_core.py:890-916
    D = 600.0
    p = np.array([900.0, 1100.0, 1500.0])
    f_o = np.array([5.0, 10.0, 15.0])
    f_v, h_v = pio.oblique_to_vertical(D, p, f_o)
    Re_val = library.constants()[2]
    dcurv = Re_val * (1.0 - np.cos((D / 2.0) / Re_val))
    phi = np.arcsin(D / p)
    assert np.allclose(f_v / f_o, np.cos(phi), rtol=1e-12, atol=1e-12)
    assert np.allclose(h_v + dcurv, 0.5 * p * np.cos(phi), rtol=1e-12, atol=1e-12)
    assert np.all(np.isfinite(f_v)) and np.all(np.isfinite(h_v)) and np.all(h_v >= 0)


def test_input_dict_adapter_single_and_batch():
    g = load_golden("g4_day_night.npz")
    day = {k: g[f"Day_{k}"] for k in ("alt", "den", "bmag", "bpsi")}
    day.update(year=2025, month=9, day=1, UT=0, F107=204, tlat=4.5, tlon=-150.0)     # generate_input_1D extras
    night = {k: g[f"Night_{k}"][None, :] for k in ("alt", "den", "bmag", "bpsi")}    # un-squeezed arrays are accepted
    alt, den, bmag, bpsi = pio.from_input_dict(day)
    assert alt.shape == den.shape == (620,) and den.dtype == np.float64
    alt2, den2, bmag2, bpsi2 = pio.from_input_dict([day, night])
    assert alt2.shape == (620,) and den2.shape == bmag2.shape == bpsi2.shape == (2, 620)
    assert np.array_equal(den2[1], g["Night_den"])
    with pytest.raises(KeyError):
        pio.from_input_dict({"alt": day["alt"]})
    with pytest.raises(ValueError):
        pio.from_input_dict({**day, "den": day["den"][:10]})


def test_batch_npz_round_trip(tmp_path):
    g = load_golden("g5_chapman64.npz")
    path = tmp_path / "batch.npz"
    pio.save_batch_npz(path, g["alt"], g["den"][:5], g["bmag"][:5], g["bpsi"][:5], freq=g["freq"], seed=20260001)
    back = pio.load_batch_npz(path)
    for k in ("alt", "freq"):
        assert np.array_equal(back[k], g[k])
    for k in ("den", "bmag", "bpsi"):
        assert np.array_equal(back[k], g[k][:5])
    assert int(back["meta"]["seed"]) == 20260001
    with pytest.raises(ValueError):
        pio.save_batch_npz(path, g["alt"][:-1], g["den"][:5], g["bmag"][:5], g["bpsi"][:5])


def test_tracer_helpers_like_the_reference():
    """tan_from_mu_scalar and find_turning_point (reference library.py:1034-1093) - the checks of the reference's
    test_core.py:613-635 plus the bracket rules of its loop."""
    from pyrayhf_amd.tracers import find_turning_point, tan_from_mu_scalar
    np.testing.assert_allclose(tan_from_mu_scalar(2.0, 1.0), 1.0 / np.sqrt(3.0), rtol=1e-12)
    near = tan_from_mu_scalar(1.0000001, 1.0)
    assert np.isfinite(near) and near > 0.0
    assert tan_from_mu_scalar(1.0, 1.0) == 1.0 / np.sqrt(1e-10)          # the 1e-10 floor under the root
    small = tan_from_mu_scalar(1e-6, 1e-7)
    assert np.isfinite(small) and small >= 0.0
    assert tan_from_mu_scalar(np.array([2.0, 3.0]), 1.0).shape == (2,)
    z = np.array([100.0, 110.0, 120.0, 130.0])
    assert find_turning_point(z, np.array([1.0, 0.8, 0.4, 0.2]), 0.6) == 115.0
    assert find_turning_point(z, np.array([1.0, 0.6, 0.6, 0.2]), 0.6) == 100.0 + (0.4 / 0.4) * 10.0   # first bracket
    assert find_turning_point(z, np.array([0.6, 0.6, 0.4, 0.2]), 0.6) == 100.0                       # equal ends: the lower node
    assert np.isnan(find_turning_point(z, np.array([1.0, 0.9, 0.8, 0.7]), 0.6))
