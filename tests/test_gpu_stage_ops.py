"""The un-fused stage functions of the path as standalone device ops, against the reference's
stage captures (fixture G6, reference regrid_to_nonuniform_grid / find_X / find_Y / find_mu_mup /
find_vh run on the Day profile) and its structural tests."""

import numpy as np
import pytest

from conftest import load_golden, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pyrayhf_amd import library
    return library


@pytest.mark.parametrize("mode", ["O", "X"])
def test_regrid_is_bit_identical_to_the_reference(lib, mode):
    g = load_golden("g4_day_night.npz")
    s = load_golden("g6_stages.npz")
    rg = lib.regrid_to_nonuniform_grid(s["freq"] * 1e6, g["Day_den"], g["Day_bmag"], g["Day_bpsi"], g["Day_alt"],
                                       mode=mode, n_points=50)
    assert set(rg) == {"freq", "den", "bmag", "bpsi", "dist", "alt", "crit_height", "ind"}   # library.py:430-437
    for key in ("den", "bmag", "bpsi", "dist", "alt", "crit_height"):
        assert rg[key].shape == (3, 50)
        assert same_bits(rg[key], s[f"{mode}_{key}"]), key
    assert np.array_equal(rg["freq"], np.repeat(s["freq"][:, None] * 1e6, 50, axis=1))
    assert rg["ind"].dtype == np.int64 and np.array_equal(rg["ind"], np.tile(np.arange(50), (3, 1)))


def test_regrid_basic_structure_and_escaping_rows(lib):
    # reference test_core.py:191-207
    f = np.array([1.0e6, 2.0e6, 2.0e7])
    rg = lib.regrid_to_nonuniform_grid(f, np.array([1.0e11, 5.0e11, 1.0e12]), np.full(3, 5.0e-5), np.full(3, 60.0),
                                       np.array([100, 200, 300]), mode="O", n_points=10)
    assert isinstance(rg, dict) and rg["freq"].shape[0] == len(f) and rg["den"].shape[0] == len(f)
    # 20 MHz escapes: NaN altitudes, but the last thickness is still the 1e-6 km back-off (library.py:415-416)
    assert np.all(np.isnan(rg["alt"][2])) and np.all(np.isnan(rg["dist"][2, :-1])) and rg["dist"][2, -1] == 1e-6
    assert np.all(np.isnan(rg["crit_height"][2]))
    with pytest.raises(ValueError, match="mode must be 'O' or 'X'"):
        lib.regrid_to_nonuniform_grid(f, np.ones(3), np.ones(3), np.ones(3), np.arange(3.0), mode="Q")


@pytest.mark.parametrize("mode", ["O", "X"])
def test_find_vh_on_stage_captures(lib, mode):
    g = load_golden("g4_day_night.npz")
    s = load_golden("g6_stages.npz")
    vh = lib.find_vh(s[f"{mode}_X"], s[f"{mode}_Y"], s[f"{mode}_bpsi"], s[f"{mode}_dist"],
                     float(np.min(g["Day_alt"])), mode)
    assert vh.shape == (3,)
    np.testing.assert_allclose(vh, s[f"{mode}_vh"], rtol=1e-6 if mode == "O" else 1e-11)


def test_find_vh_small_known_answer(lib):
    # reference test_core.py:155-168 (+ the value the reference returns for it)
    k = load_golden("g3_index_kat.npz")
    vh = lib.find_vh(np.array([[0.5, 0.6]]), np.array([[0.1, 0.2]]), np.array([[45.0, 45.0]]),
                     np.array([[1.0, 1.0]]), 100.0, "O")
    assert isinstance(vh, np.ndarray) and vh.shape == (1,) and vh[0] > 100.0
    np.testing.assert_allclose(vh, k["find_vh_small"], rtol=1e-13)
    # every term NaN -> exact zero -> NaN (library.py:290)
    assert np.isnan(lib.find_vh(np.array([[1.5, 2.0]]), np.array([[0.1, 0.2]]), np.array([[45.0, 45.0]]),
                                np.array([[1.0, 1.0]]), 100.0, "O")[0])


def test_unfused_chain_equals_fused_operator(lib):
    """regrid -> find_X / find_Y -> find_vh (the reference's own composition, library.py:495-507)
    against the fused kernel on the same profile."""
    g = load_golden("g4_day_night.npz")
    freq = g["freq"][20:120:7]
    args = (g["Night_den"], g["Night_bmag"], g["Night_bpsi"], g["Night_alt"])
    for mode, n in (("X", 400), ("O", 200)):
        rg = lib.regrid_to_nonuniform_grid(freq * 1e6, *args, mode=mode, n_points=n)
        with np.errstate(all="ignore"):
            X = lib.find_X(rg["den"], rg["freq"])
            Y = lib.find_Y(rg["freq"], rg["bmag"])
        chain = lib.find_vh(X, Y, rg["bpsi"], rg["dist"], float(np.min(g["Night_alt"])), mode)
        fused = lib.vertical_forward_operator(freq, *args, mode, n, math=lib.MATH_FAITHFUL)
        assert np.array_equal(np.isnan(chain), np.isnan(fused))
        ok = np.isfinite(fused)
        np.testing.assert_allclose(chain[ok], fused[ok], rtol=1e-12 if mode == "X" else 2e-5)
