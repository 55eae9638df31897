"""The plain-C restatement (oracle/vfo_oracle.c) against the reference's golden vectors.

It is not bit-identical to the NumPy path (libm vs NumPy SIMD sin/cos/pow), so it is held to
the same parity rule as the HIP kernel."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import vfo_c
from parity import assert_masks, assert_o_mode, assert_x_mode, rel_err



@pytest.fixture(scope="module", autouse=True)
def _checker_is_there():
    vfo_c.require()        # built from oracle/vfo_oracle.c when absent; a missing checker fails, it does not skip


def test_basic_and_edp_known_answers():
    g = load_golden("g1_basic.npz")
    assert_o_mode(vfo_c.virtual_heights_batch(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 50), g["vh_O"])
    assert_x_mode(vfo_c.virtual_heights_batch(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 50), g["vh_X"],
                  tol=1e-12)
    g = load_golden("g2_edp_kat.npz")
    vh = vfo_c.virtual_heights_batch(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"])
    np.testing.assert_allclose(vh, g["vh_published"], rtol=1e-6)        # reference test_core.py:275


@pytest.mark.parametrize("which", ["Day", "Night"])
def test_day_night(which):
    g = load_golden("g4_day_night.npz")
    args = (g["freq"], g[f"{which}_den"], g[f"{which}_bmag"], g[f"{which}_bpsi"], g[f"{which}_alt"])
    for n in (200, 2000):
        assert_x_mode(vfo_c.virtual_heights_batch(*args, "X", n), g[f"{which}_X_{n}_vh"], tol=1e-11)
        assert_o_mode(vfo_c.virtual_heights_batch(*args, "O", n), g[f"{which}_O_{n}_vh"], g[f"{which}_O_{n}_noise"])


def test_chapman_batch_and_threads():
    g = load_golden("g5_chapman64.npz")
    vx = vfo_c.virtual_heights_batch(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 2000)
    assert_x_mode(vx, g["X_2000_vh"], tol=1e-10)
    vo = vfo_c.virtual_heights_batch(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 200)
    assert_o_mode(vo, g["O_200_vh"], g["O_200_noise"])
    one = vfo_c.virtual_heights_batch(g["freq"], g["den"][:4], g["bmag"][:4], g["bpsi"][:4], g["alt"], "O", 200,
                                      n_threads=1)
    assert np.array_equal(one, vo[:4], equal_nan=True)          # threading does not change results


def test_edge_cases():
    g = load_golden("g7_edges.npz")
    names = sorted({k[: -len("_n_points")] for k in g if k.endswith("_n_points")})
    for name in names:
        args = [g[f"{name}_{k}"] for k in ("freq", "den", "bmag", "bpsi", "alt")]
        for mode in "OX":
            vh = vfo_c.virtual_heights_batch(*args, mode, int(g[f"{name}_n_points"]))
            want = g[f"{name}_vh_{mode}"]
            assert_masks(vh, want)
            err, ok = rel_err(vh, want)
            assert err.max(initial=0.0) <= 2e-6, (name, mode, err.max())


def test_config5_rows_g15():
    """The reference-run rows at config 5's shapes (512 freqs; O/200, X/2000, O/2000, X/20000)."""
    g = load_golden("g15_config5_rows.npz")
    for mode, n, tol in (("X", 2000, 1e-11), ("X", 20000, 1e-11)):
        got = vfo_c.virtual_heights_batch(g["freq"], g[f"X_{n}_den"], g[f"X_{n}_bmag"], g[f"X_{n}_bpsi"], g["alt"], "X", n)
        assert_x_mode(got, g[f"X_{n}_vh"], tol=tol)
    for n in (200, 2000):
        got = vfo_c.virtual_heights_batch(g["freq"], g[f"O_{n}_den"], g[f"O_{n}_bmag"], g[f"O_{n}_bpsi"], g["alt"], "O", n)
        assert_o_mode(got, g[f"O_{n}_vh"], np.maximum(g[f"O_{n}_noise"], g[f"O_{n}_noise_rounding"]))
