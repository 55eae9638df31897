"""The plain-C restatement (oracle/vfo_oracle.c) against the reference's golden vectors.

It is not bit-identical to the NumPy path (libm vs NumPy SIMD sin/cos/pow), so it is held to
the same parity rule as the HIP kernel."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import vfo_c
from parity import assert_masks, assert_o_mode, assert_x_mode, rel_err

pytestmark = pytest.mark.skipif(not vfo_c.available(), reason="oracle/libvfo_oracle.so not built (make -C oracle)")


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
