"""GPU parity tests proper: the HIP path (through the C ABI) against the committed golden
vectors of the reference and against the oracle on seeded inputs.  Run with -m gpu."""

import numpy as np
import pytest

from conftest import load_golden
from parity import (assert_masks, assert_o_mode, assert_o_mode_reference_noise_alone, assert_x_mode, combined_noise,
                    oracle_noise, rel_err)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pyrayhf_amd import library
    return library


def test_basic_operator_g1(lib):
    g = load_golden("g1_basic.npz")
    # reference test_core.py:223-236 (alt is an int array there)
    vh = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"].astype(int),
                                       mode="O", n_points=int(g["n_points"]))
    assert isinstance(vh, np.ndarray) and vh.shape == g["freq"].shape
    assert np.isnan(vh[-1]) and np.all(np.isfinite(vh[:-1]))
    assert_o_mode(vh, g["vh_O"])
    vx = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", int(g["n_points"]))
    assert_x_mode(vx, g["vh_X"])


def test_edp_known_answer_g2(lib):
    g = load_golden("g2_edp_kat.npz")
    vh = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"])   # defaults O/200
    np.testing.assert_allclose(vh, g["vh_published"], rtol=1e-6)      # reference test_core.py:275
    assert_o_mode(vh, g["vh_O"])


def test_scalar_frequency_gives_shape_1(lib):
    g = load_golden("g1_basic.npz")
    vh = lib.vertical_forward_operator(np.float64(2.0), g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 50)
    assert vh.shape == (1,)
    assert abs(vh[0] - g["vh_O"][1]) <= 1e-6 * g["vh_O"][1]


@pytest.mark.parametrize("which", ["Day", "Night"])
@pytest.mark.parametrize("n_points", [200, 2000, 20000])
def test_day_night_x_mode_g4(lib, which, n_points):
    g = load_golden("g4_day_night.npz")
    vh = lib.vertical_forward_operator(g["freq"], g[f"{which}_den"], g[f"{which}_bmag"], g[f"{which}_bpsi"],
                                       g[f"{which}_alt"], "X", n_points)
    worst = assert_x_mode(vh, g[f"{which}_X_{n_points}_vh"])
    print(f"{which} X n={n_points}: max rel err {worst:.3e}")


@pytest.mark.parametrize("which", ["Day", "Night"])
@pytest.mark.parametrize("n_points", [200, 2000, 20000])
def test_day_night_o_mode_g4(lib, which, n_points):
    g = load_golden("g4_day_night.npz")
    vh = lib.vertical_forward_operator(g["freq"], g[f"{which}_den"], g[f"{which}_bmag"], g[f"{which}_bpsi"],
                                       g[f"{which}_alt"], "O", n_points)
    want = g[f"{which}_O_{n_points}_vh"]
    noise = combined_noise(g[f"{which}_O_{n_points}_noise"], load_golden("g12_rounding_noise.npz")[f"g4_{which}_O_{n_points}"])
    worst = assert_o_mode(vh, want, noise)
    err, ok = rel_err(vh, want)
    print(f"{which} O n={n_points}: max rel err {worst:.3e}; worst err/noise "
          f"{np.max(err[ok] / np.maximum(noise[ok], 1e-16)):.2f}; within 1e-6: {(err[ok] <= 1e-6).mean():.3f}")


def test_chapman_batch_g5(lib):
    g = load_golden("g5_chapman64.npz")
    vo = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 200)
    assert vo.shape == (64, g["freq"].size)
    assert_o_mode(vo, g["O_200_vh"], combined_noise(g["O_200_noise"], load_golden("g12_rounding_noise.npz")["g5_O_200"]))
    vx = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 2000)
    assert_x_mode(vx, g["X_2000_vh"])
    # batch == loop of single-profile calls.  Not bit for bit: a lone profile is cut into chunks (another
    # summation order) and, with 174 pairs on a 2000-point grid, skips the pair table and runs the generic
    # loop (exact segment search) where the batch runs the main loop (closed-form segment index): the two
    # evaluations agree to ~1e-12, thirty times below the reference's own +-1 ulp response in X mode (3e-11)
    one = lib.vertical_forward_operator(g["freq"], g["den"][5], g["bmag"][5], g["bpsi"][5], g["alt"], "X", 2000)
    assert_x_mode(one, vx[5], tol=1e-11)


def test_config3_rows_o_mode_g10(lib):
    """The configuration the O-mode tolerance is about (BASELINE config 3: seed 20260003, O mode, n_points = 200),
    first 64 profiles, against the reference's own output and noise floor - per pair, no frequency window."""
    g = load_golden("g10_config3_rows.npz")
    noise = combined_noise(g["O_200_noise"], load_golden("g12_rounding_noise.npz")["g10_O_200"])
    for math in (None, lib.MATH_FAITHFUL):
        vo = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 200, math=math)
        worst = assert_o_mode(vo, g["O_200_vh"], noise, min_within=0.995)
        err, ok = rel_err(vo, g["O_200_vh"])
        print(f"G10 math={math}: within 1e-6 {(err[ok] <= 1e-6).mean():.4f}, bit-identical {(vo[ok] == g['O_200_vh'][ok]).mean():.3f}, "
              f"worst {worst:.2e}")


def test_config3_rows_against_the_reference_noise_alone_g10(lib):
    """The same rows under SURVEY 8(d)'s rule as written - |gpu - ref| <= max(1e-6, 4 x the REFERENCE's recorded
    input-jitter noise), nothing made with the oracle in the floor - as a count: the two pairs that only the
    rounding noise explains (NumPy's pow is one ulp off there, DESIGN.md section 2: 1.4e-6 and 4.2e-6 against an
    input noise of 1e-11) are the only ones allowed beyond it, and they stay below 1e-5.  A regression of the
    exactly rounded sin / cos / pow path cannot hide behind an oracle-made floor here."""
    g = load_golden("g10_config3_rows.npz")
    want = g["O_200_vh"]
    limit = np.minimum(1e-3, np.maximum(1e-6, 4.0 * np.where(np.isfinite(g["O_200_noise"]), g["O_200_noise"], np.inf)))
    for math in (None, lib.MATH_FAITHFUL):
        vo = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 200, math=math)
        assert_masks(vo, want)
        err, ok = rel_err(vo, want)
        beyond = ok & (err > limit)
        assert int(beyond.sum()) <= 2 and err[beyond].max(initial=0.0) <= 1e-5, (math, int(beyond.sum()), err[beyond])
        assert (err[ok] <= 1e-6).mean() >= 0.998, (math, (err[ok] <= 1e-6).mean())


@pytest.mark.parametrize("which", ["Day", "Night"])
@pytest.mark.parametrize("n_points", [200, 2000, 20000])
def test_day_night_against_the_reference_noise_alone_g4(lib, which, n_points):
    """G4's O-mode rows under SURVEY 8(d)'s rule as written (the reference's recorded noise alone in the floor), as a
    count: NO pair beyond it, in the default and in the reference-order arithmetic (measured: worst pair 1.5e-6 on
    Day, inside 4 x its recorded noise; 3.7e-8 ... 1.7e-7 on Night)."""
    g = load_golden("g4_day_night.npz")
    for math in (None, lib.MATH_FAITHFUL):
        vh = lib.vertical_forward_operator(g["freq"], g[f"{which}_den"], g[f"{which}_bmag"], g[f"{which}_bpsi"],
                                           g[f"{which}_alt"], "O", n_points, math=math)
        assert_o_mode_reference_noise_alone(vh, g[f"{which}_O_{n_points}_vh"], g[f"{which}_O_{n_points}_noise"],
                                            allowed_beyond=0, min_within=0.99)


def test_chapman_batch_against_the_reference_noise_alone_g5(lib):
    """G5's O/200 rows (64 seeded Chapman profiles, 5 584 finite pairs): no pair beyond the reference's own noise
    rule (measured: worst 3.9e-6, inside 4 x its recorded noise), 99.9 % within 1e-6 outright."""
    g = load_golden("g5_chapman64.npz")
    for math in (None, lib.MATH_FAITHFUL):
        vo = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 200, math=math)
        assert_o_mode_reference_noise_alone(vo, g["O_200_vh"], g["O_200_noise"], allowed_beyond=0, min_within=0.999)


@pytest.mark.parametrize("math", ["faithful", "fast"])
def test_both_tiers_both_modes(lib, math):
    """Every (tier, mode) combination against the reference vectors, not just the defaults."""
    g = load_golden("g5_chapman64.npz")
    level = lib.MATH_FAITHFUL if math == "faithful" else lib.MATH_FAST
    vx = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 2000, math=level)
    print(math, "X max rel err", assert_x_mode(vx, g["X_2000_vh"]))
    vo = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 200, math=level)
    err, ok = rel_err(vo, g["O_200_vh"])
    print(math, "O max rel err", err.max(), "within 1e-6:", (err[ok] <= 1e-6).mean())
    if math == "faithful":
        assert_o_mode(vo, g["O_200_vh"], combined_noise(g["O_200_noise"], load_golden("g12_rounding_noise.npz")["g5_O_200"]))
    else:
        # The reduced algebra everywhere is NOT parity grade in O mode (opt-in, never the default there): it
        # re-associates D, so it differs from the reference by the reference's own conditioning, not by a bit of
        # sin or pow.  Statistical statement only: masks identical, median far below 1e-6, 97 % within 1e-6.
        assert_masks(vo, g["O_200_vh"])
        assert np.median(err[ok]) <= 1e-7 and (err[ok] <= 1e-6).mean() >= 0.97 and err.max() <= 1e-3


@pytest.mark.parametrize("math", ["faithful", "fast"])
def test_find_mu_mup_device_op(lib, math):
    level = lib.MATH_FAITHFUL if math == "faithful" else lib.MATH_FAST
    g = load_golden("g3_index_kat.npz")
    mu, mup = lib.find_mu_mup(g["X"], g["Y"], g["psi"], "O", math=level)
    np.testing.assert_allclose(mu, g["mu_published"], rtol=1e-5)       # reference test_core.py:137-152
    np.testing.assert_allclose(mup, g["mup_published"], rtol=1e-5)
    np.testing.assert_allclose(mu, g["mu_O"], rtol=1e-12)
    np.testing.assert_allclose(mup, g["mup_O"], rtol=1e-10)
    mu, mup = lib.find_mu_mup(g["X"], g["Y"], g["psi"], "X", math=level)
    assert np.array_equal(np.isnan(mu), np.isnan(g["mu_X"])) and np.array_equal(np.isnan(mup), np.isnan(g["mup_X"]))
    ok = np.isfinite(g["mu_X"])
    np.testing.assert_allclose(mu[ok], g["mu_X"][ok], rtol=1e-12)
    np.testing.assert_allclose(mup[ok], g["mup_X"][ok], rtol=1e-10)
    mu, mup = lib.find_mu_mup(g["unmag_X"], np.zeros(3), g["psi"], "O", math=level)    # isotropic branch
    assert np.array_equal(np.isnan(mu), np.isnan(g["unmag_mu"]))
    np.testing.assert_allclose(mu[:1], g["unmag_mu"][:1], rtol=1e-15)
    np.testing.assert_allclose(mup[:1], g["unmag_mup"][:1], rtol=1e-15)
    with pytest.raises(ValueError, match="Mode must be O or X"):
        lib.find_mu_mup(g["X"], g["Y"], g["psi"], "Z")
    # 2-D stage captures of the Day profile (fixture G6): X-mode is well conditioned everywhere
    s = load_golden("g6_stages.npz")
    mu, mup = lib.find_mu_mup(s["X_X"], s["X_Y"], s["X_bpsi"], "X", math=level)
    assert mu.shape == s["X_mu"].shape
    ok = np.isfinite(s["X_mup"])
    assert np.array_equal(np.isnan(mup), ~ok)
    well = ok & (s["X_mu"] > 0.05)
    np.testing.assert_allclose(mu[well], s["X_mu"][well], rtol=1e-9)
    np.testing.assert_allclose(mup[well], s["X_mup"][well], rtol=1e-7)
    # towards reflection mu -> 0 and both indices are conditioned like 1/mu^2 (mu ~ 1e-4 at the last point)
    np.testing.assert_allclose(mu[ok], s["X_mu"][ok], rtol=1e-5)
    np.testing.assert_allclose(mup[ok], s["X_mup"][ok], rtol=1e-4)


def test_field_angle_jump_mixes_cubic_and_trig_segments(lib):
    """A profile whose field angle jumps at two levels: those segments take the sincos path, the
    others the per-segment sin^2 cubic; the fast tier must still agree with the faithful one."""
    g = load_golden("g5_chapman64.npz")
    bpsi = g["bpsi"][:6].copy()
    bpsi[:, 40:] += 1.5                  # 1.5 degree step between levels 39 and 40
    bpsi[:, 120:] -= 0.7
    for alt in (g["alt"], np.cumsum(np.where(np.arange(g["alt"].size) % 7 == 0, 1.3, 1.0)) + 79.0):
        slow = lib.vertical_forward_operator(g["freq"], g["den"][:6], g["bmag"][:6], bpsi, alt, "X", 2000,
                                             math=lib.MATH_FAITHFUL)
        fast = lib.vertical_forward_operator(g["freq"], g["den"][:6], g["bmag"][:6], bpsi, alt, "X", 2000,
                                             math=lib.MATH_FAST)
        assert_x_mode(fast, slow, tol=1e-9)
        assert np.isfinite(fast).mean() > 0.3


def test_edge_cases_g7(lib):
    g = load_golden("g7_edges.npz")
    names = sorted({k[: -len("_n_points")] for k in g if k.endswith("_n_points")})
    for name in names:
        args = [g[f"{name}_{k}"] for k in ("freq", "den", "bmag", "bpsi", "alt")]
        n = int(g[f"{name}_n_points"])
        for mode in "OX":
            vh = lib.vertical_forward_operator(*args, mode, n)
            want = g[f"{name}_vh_{mode}"]
            if mode == "X":
                # vh ~ alt_min + 1e-6 km in the clamped cases: the sum itself is only 1e-6 of the result
                worst = assert_x_mode(vh, want, tol=1e-9)
            else:
                worst = assert_o_mode(vh, want, oracle_noise(*args, "O", n, runs=24))
            print(f"G7 {name} {mode}: worst {worst:.2e}")


def test_error_behaviour(lib):
    g = load_golden("g1_basic.npz")
    args = (g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"])
    with pytest.raises(ValueError, match="mode must be 'O' or 'X'"):
        lib.vertical_forward_operator(*args, "Z", 10)
    neg = g["den"].copy()
    neg[1] = -1.0
    with pytest.raises(ValueError, match="Density must be non-negative"):
        lib.vertical_forward_operator(g["freq"], neg, g["bmag"], g["bpsi"], g["alt"], "O", 10)
    with pytest.raises(IndexError):
        lib.vertical_forward_operator(g["freq"], g["den"][::-1].copy(), g["bmag"], g["bpsi"], g["alt"], "O", 10)
    with pytest.raises(ValueError):
        lib.vertical_forward_operator(g["freq"], g["den"][:2], g["bmag"], g["bpsi"], g["alt"], "O", 10)
    # a frequency that is not a positive finite number: a NaN column and the other columns untouched, NumPy and
    # GPU-resident inputs alike (the reference: NaN for 0 and NaN - a sweep padded with NaN works there too - and
    # something meaningless for a negative frequency)
    import torch
    clean = lib.vertical_forward_operator(np.array([2.0, 3.0]), g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 10)
    for mode, n in (("O", 10), ("X", 10), ("O", 200), ("X", 2000)):
        clean = lib.vertical_forward_operator(np.array([2.0, 3.0]), g["den"], g["bmag"], g["bpsi"], g["alt"], mode, n)
        for bad_f in (0.0, -3.0, np.nan, np.inf):
            f = np.array([2.0, bad_f, 3.0])
            vh = lib.vertical_forward_operator(f, g["den"], g["bmag"], g["bpsi"], g["alt"], mode, n)
            assert np.isnan(vh[1]) and np.array_equal(vh[[0, 2]], clean, equal_nan=True), (mode, n, bad_f, vh)
            t = [torch.as_tensor(x, device="cuda:0") for x in (f, g["den"], g["bmag"], g["bpsi"], g["alt"].astype(float))]
            vt = lib.vertical_forward_operator(*t, mode, n).cpu().numpy()
            assert np.array_equal(vt, vh, equal_nan=True), (mode, n, bad_f)
    # ... also in a batch that takes the long-launch paths (candidate list, short-grid kernel)
    g5 = load_golden("g5_chapman64.npz")
    f = g5["freq"].copy()
    f[[7, 60]] = [np.nan, -1.0]
    for mode, n in (("O", 200), ("X", 2000)):
        vh = lib.vertical_forward_operator(f, g5["den"], g5["bmag"], g5["bpsi"], g5["alt"], mode, n)
        want = lib.vertical_forward_operator(g5["freq"], g5["den"], g5["bmag"], g5["bpsi"], g5["alt"], mode, n)
        keep = np.ones(f.size, bool)
        keep[[7, 60]] = False
        assert np.isnan(vh[:, ~keep]).all() and np.array_equal(vh[:, keep], want[:, keep], equal_nan=True), mode
    # NaN in a profile is not an error: the result is what the reference computes (fixture G13 `nanfield`; here in a
    # batch, where the NaN profile leaves the short-grid kernels for the general one and the others must not notice)
    from oracle import vfo_numpy as orc_nan
    for col in ("alt", "bmag", "bpsi"):
        for n, mode in ((10, "O"), (200, "O"), (200, "X"), (2000, "X")):
            bad = {k: g5[k].copy() for k in ("den", "bmag", "bpsi")}
            alt5 = np.tile(g5["alt"], (64, 1))
            (alt5 if col == "alt" else bad[col])[11, 3] = np.nan         # level 3: below every peak
            got = lib.vertical_forward_operator(g5["freq"], bad["den"], bad["bmag"], bad["bpsi"], alt5, mode, n)
            clean = lib.vertical_forward_operator(g5["freq"], g5["den"], g5["bmag"], g5["bpsi"], g5["alt"], mode, n)
            rows = np.arange(64) != 11
            assert np.array_equal(got[rows], clean[rows], equal_nan=True), (col, mode, n)
            with np.errstate(all="ignore"):
                want11 = orc_nan.virtual_heights(g5["freq"], bad["den"][11], bad["bmag"][11], bad["bpsi"][11], alt5[11], mode, n)
            assert_masks(got[11], want11)
            err, ok = rel_err(got[11], want11)
            assert err.max(initial=0.0) <= (1e-8 if mode == "X" else 1e-5), (col, mode, n, err.max())
    # ... except in the density: np.argmax ranks a NaN as the maximum (library.py:371), so the column is cut there
    from oracle import vfo_numpy as orc
    cut = g5["den"].copy()
    cut[11, 40:] = np.nan                                                # profile 11: NaN from level 40 up
    for n, mode in ((200, "O"), (2000, "X")):
        got = lib.vertical_forward_operator(g5["freq"], cut, g5["bmag"], g5["bpsi"], g5["alt"], mode, n)
        with np.errstate(all="ignore"):
            want11 = orc.virtual_heights(g5["freq"], cut[11], g5["bmag"][11], g5["bpsi"][11], g5["alt"], mode, n)
        clean = lib.vertical_forward_operator(g5["freq"], g5["den"], g5["bmag"], g5["bpsi"], g5["alt"], mode, n)
        rows = np.arange(64) != 11
        assert np.array_equal(got[rows], clean[rows], equal_nan=True)
        assert_masks(got[11], want11)
        err, ok = rel_err(got[11], want11)
        assert err.max(initial=0.0) <= (1e-8 if mode == "X" else 1e-5), (mode, err.max())
    top = g5["bmag"].copy()
    top[:, -1] = np.nan                                                  # above the peak: never read, as in the reference
    ok_vh = lib.vertical_forward_operator(g5["freq"], g5["den"], top, g5["bpsi"], g5["alt"], "X", 2000)
    assert np.array_equal(ok_vh, lib.vertical_forward_operator(g5["freq"], g5["den"], g5["bmag"], g5["bpsi"], g5["alt"], "X", 2000),
                          equal_nan=True)
    # the context stays usable after a data error
    vh = lib.vertical_forward_operator(*args, "O", 50)
    assert np.isfinite(vh[0])


def test_inputs_are_not_mutated(lib):
    g = load_golden("g1_basic.npz")
    args = [g[k].copy() for k in ("freq", "den", "bmag", "bpsi", "alt")]
    before = [a.copy() for a in args]
    lib.vertical_forward_operator(*args, "X", 50)
    for a, b in zip(args, before):
        assert np.array_equal(a, b)


def test_oracle_on_fresh_seeded_batch(lib):
    """Seeded inputs never seen by the fixtures: HIP vs the oracle at oracle-friendly sizes."""
    from oracle import vfo_numpy as orc
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(24, 4242)
    freq = synth.sounder_frequencies(4)
    got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", 2000)
    want = orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", 2000)
    assert_x_mode(got, want)
    got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", 200)
    want = orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "O", 200)
    assert_o_mode(got, want, oracle_noise(freq, den, bmag, bpsi, alt, "O", 200))


@pytest.mark.parametrize("n_points", [1, 200, 20000])
def test_x_mode_below_the_gyrofrequency_grid_collapses_onto_the_bottom_level(lib, n_points):
    """f < f_H at the bottom level: X + Y > 1 there, the reference's reflection height is alt[0] - 1e-6 and all of
    its grid is clamped to level 0 (library.py:399-416) - the kernel answers those pairs without a loop
    (collapsed_grid_sum).  Same masks and values as the oracle, batch launch and single-profile (chunked) launch;
    an altitude grid that starts at 0 km (where the reference's sum cancels to nothing) keeps the generic loop."""
    from oracle import vfo_numpy as orc
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(12, 777)
    freq = np.linspace(0.3, 2.6, 47)                     # f_H at 80 km: 0.6 ... 1.7 MHz
    want = orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", n_points)
    collapsed = np.isfinite(want) & (want - alt.min() < 1e-9)
    if n_points > 1:
        assert collapsed.any() and np.isnan(want).any() and (np.isfinite(want) & ~collapsed).any()
    got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", n_points)
    assert_x_mode(got, want)
    one = lib.vertical_forward_operator(freq, den[3], bmag[3], bpsi[3], alt, "X", n_points)
    assert_x_mode(one, want[3])
    if n_points == 200:
        alt0 = alt - alt[0]
        want0 = orc.virtual_heights_batch(freq, den, bmag, bpsi, alt0, "X", n_points)
        got0 = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt0, "X", n_points)
        ok = np.isfinite(want0) & (want0 > 1e-9) & np.isfinite(got0)
        assert ok.any() and np.max(np.abs(got0[ok] - want0[ok]) / want0[ok]) <= 1e-9
        # (the pairs whose sum cancels to +-1e-22 or exactly 0 in the reference are its own coin flips: not compared)
        assert np.array_equal(np.isnan(got0[want0 > 1e-9]), np.isnan(want0[want0 > 1e-9]))


@pytest.mark.parametrize("n_points", [200, 2000])
def test_o_mode_below_the_bottom_plasma_frequency_grid_collapses_onto_the_bottom_level(lib, n_points):
    """O mode, f below the plasma frequency of the lowest level: X >= 1 there, the same collapsed grid as in X mode
    below the gyrofrequency - answered without a loop in the default arithmetic (short grids: the four-frequency
    items; long grids: the per-pair items) and in the reference order."""
    from oracle import vfo_numpy as orc
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(48, 778)
    den = den + np.linspace(1.0e9, 6.0e9, 48)[:, None] * np.exp(-(alt - 80.0) / 400.0)   # f_N(80 km) = 0.28 ... 0.70 MHz
    freq = np.linspace(0.2, 3.0, 90)
    want = orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "O", n_points)
    # (in O mode mu^2 < 0 at X > 1, library.py:233: every term of a collapsed grid is NaN and so is the trace)
    below_bottom = (freq[None, :] * 1e6) ** 2 < 80.6163849 * den[:, :1]
    assert (below_bottom & np.isnan(want)).sum() > 50 and np.isfinite(want).sum() > 1000
    assert not np.isfinite(want[below_bottom]).any()
    noise = oracle_noise(freq, den, bmag, bpsi, alt, "O", n_points)
    for math in (None, lib.MATH_FAITHFUL):
        got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", n_points, math=math)
        assert_o_mode(got, want, noise)                      # NaN masks identical, values by the noise rule


def test_x_mode_heights_per_thread_with_ties_plateaus_and_no_field(lib):
    """A batch big enough for the prologue's per-thread level scan in X mode (>= 4096 pairs, n_freq x n_points >=
    1e6): plateaus under a uniform field (exact ties of X + Y between levels: the scan repeats with exact
    divisions), no field at all, ordinary rows - against the NumPy oracle, and against single-profile launches,
    which scan per pair."""
    from oracle import vfo_numpy as orc
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(40, 4321)
    bmag[:10] = bmag[:10, :1]                                   # uniform field ...
    for r in range(10):
        k = 40 + 7 * r
        den[r, k:k + 3] = den[r, k]                             # ... and a plateau: equal X + Y at three levels
    den[5:10, :30] = 0.0                                        # vacuum at the bottom as well
    bmag[10:20] = 0.0                                           # isotropic rows
    freq = np.linspace(0.3, 14.0, 128)
    n_points = 8192
    want = orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", n_points)
    got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", n_points)
    assert_x_mode(got, want)
    for r in (0, 7, 12, 25):
        one = lib.vertical_forward_operator(freq, den[r], bmag[r], bpsi[r], alt, "X", n_points)
        assert_x_mode(one, got[r], tol=1e-11)


def test_between_512_and_1024_frequencies(lib):
    """A candidate list exists (<= 1024 frequencies, >= 4096 pairs) but X mode has no room for per-thread heights
    (more frequencies than threads): O mode settles its heights in the prologue, X mode scans per pair."""
    from oracle import vfo_numpy as orc
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(8, 606)
    freq = np.linspace(0.3, 15.0, 600)
    got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "X", 300)
    assert_x_mode(got, orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", 300))
    got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", 200)
    assert_o_mode(got, orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "O", 200),
                  oracle_noise(freq, den, bmag, bpsi, alt, "O", 200))


def test_per_profile_altitude_rows(lib):
    g = load_golden("g5_chapman64.npz")
    alt2 = np.tile(g["alt"], (8, 1))
    a = lib.vertical_forward_operator(g["freq"], g["den"][:8], g["bmag"][:8], g["bpsi"][:8], alt2, "X", 400)
    b = lib.vertical_forward_operator(g["freq"], g["den"][:8], g["bmag"][:8], g["bpsi"][:8], g["alt"], "X", 400)
    assert np.array_equal(a, b, equal_nan=True)


def test_mixed_worklist_matches_separate_launches(lib):
    g = load_golden("g5_chapman64.npz")
    args = (g["freq"], g["den"][:24], g["bmag"][:24], g["bpsi"][:24], g["alt"])
    segs = [(0, 8, "O", 200), (8, 14, "X", 2000), (14, 20, "O", 2000), (20, 24, "X", 20000)]
    mixed = lib.vertical_forward_operator_mixed(*args, segs)       # one launch, each slice in its own tier
    for p0, p1, mode, n in segs:
        sep = lib.vertical_forward_operator(g["freq"], g["den"][p0:p1], g["bmag"][p0:p1], g["bpsi"][p0:p1],
                                            g["alt"], mode, n)
        assert np.array_equal(mixed[p0:p1], sep, equal_nan=True), (p0, p1, mode, n)


def test_torch_device_resident_inputs(lib):
    import torch
    g = load_golden("g5_chapman64.npz")
    dev = torch.device("cuda:0")
    t = {k: torch.as_tensor(g[k], device=dev) for k in ("freq", "den", "bmag", "bpsi", "alt")}
    out = lib.vertical_forward_operator(t["freq"], t["den"], t["bmag"], t["bpsi"], t["alt"], "X", 2000)
    assert out.is_cuda and out.shape == (64, g["freq"].size)
    host = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 2000)
    assert np.array_equal(out.cpu().numpy(), host, equal_nan=True)


def test_torch_async_launch_is_ordered_on_the_callers_stream(lib):
    """sync=False enqueues on torch's current stream (default or side stream): a torch op issued right
    after the call must see the finished result without any host synchronisation in between."""
    import torch
    g = load_golden("g5_chapman64.npz")
    dev = torch.device("cuda:0")
    t = {k: torch.as_tensor(g[k], device=dev) for k in ("freq", "den", "bmag", "bpsi", "alt")}
    want = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 20000)
    for stream in (torch.cuda.current_stream(dev), torch.cuda.Stream(dev)):
        with torch.cuda.stream(stream):
            out = torch.full((64, g["freq"].size), -1.0, dtype=torch.float64, device=dev)
            lib.vertical_forward_operator(t["freq"], t["den"], t["bmag"], t["bpsi"], t["alt"], "X", 20000,
                                          sync=False, out=out)
            doubled = out * 2.0                        # same stream: must run after the kernel
        stream.synchronize()
        assert np.array_equal(doubled.cpu().numpy(), want * 2.0, equal_nan=True)
    # back to host inputs on the same context afterwards
    again = lib.vertical_forward_operator(g["freq"], g["den"][:2], g["bmag"][:2], g["bpsi"][:2], g["alt"], "X", 20000)
    # two profiles are cut into chunks: another summation order, and other grid points fall into the main loop's
    # top-segment phase (whose interpolants are the same polynomials rounded differently): agreement to ~1e-12
    assert_x_mode(again, want[:2], tol=1e-11)


def test_mixed_worklist_on_device_tensors(lib):
    import torch
    g = load_golden("g5_chapman64.npz")
    segs = [(0, 8, "O", 200), (8, 14, "X", 2000), (20, 24, "X", 20000)]          # rows 14..19 uncovered
    host = lib.vertical_forward_operator_mixed(g["freq"], g["den"][:24], g["bmag"][:24], g["bpsi"][:24], g["alt"], segs)
    dev = torch.device("cuda:0")
    t = [torch.as_tensor(x, device=dev) for x in (g["freq"], g["den"][:24], g["bmag"][:24], g["bpsi"][:24], g["alt"])]
    out = lib.vertical_forward_operator_mixed(*t, segs)
    assert out.is_cuda and out.shape == (24, g["freq"].size)
    assert np.array_equal(out.cpu().numpy(), host, equal_nan=True)
    assert np.all(np.isnan(host[14:20]))
    # sync=False: enqueued on torch's current stream; a torch op behind it sees the finished rows
    late = lib.vertical_forward_operator_mixed(*t, segs, sync=False)
    doubled = late * 2.0
    torch.cuda.synchronize(dev)
    assert np.array_equal(doubled.cpu().numpy(), host * 2.0, equal_nan=True)


def test_cached_device_grids_alternate_without_going_stale(lib):
    """GPU-resident calls reuse one device grid per n_points and tell the library it may keep the table it
    derives from it (PRHF_FLAG_GRID_STABLE).  Alternating grid sizes, a mixed work list in between and a
    host-buffer call must each get the table of their own grid."""
    import torch
    g = load_golden("g5_chapman64.npz")
    dev = torch.device("cuda:0")
    t = [torch.as_tensor(g[k], device=dev) for k in ("freq", "den", "bmag", "bpsi", "alt")]
    h = [g[k] for k in ("freq", "den", "bmag", "bpsi", "alt")]
    want = {n: lib.vertical_forward_operator(*h, "X", n) for n in (2000, 20000, 777)}
    segs = [(0, 30, "X", 777), (30, 64, "X", 2000)]
    want_mixed = lib.vertical_forward_operator_mixed(*h, segs)
    for n in (2000, 20000, 2000, 777, 20000, 777):
        got = lib.vertical_forward_operator(*t, "X", n).cpu().numpy()
        assert np.array_equal(got, want[n], equal_nan=True), n
        mixed = lib.vertical_forward_operator_mixed(*t, segs).cpu().numpy()
        assert np.array_equal(mixed, want_mixed, equal_nan=True), n
        assert np.array_equal(lib.vertical_forward_operator(*h, "X", n), want[n], equal_nan=True), n


def test_host_grid_marked_stable_is_uploaded_once_and_never_goes_stale():
    """PRHF_FLAG_GRID_STABLE with host buffers (round 3): the grid at a host address is uploaded, and its table built,
    once per (address, length).  A grid of the same length at ANOTHER address must not get the first one's copy, and
    PRHF_FLAG_ASYNC still needs device pointers."""
    from pyrayhf_amd import _native, library
    g = load_golden("g5_chapman64.npz")
    ctx = _native.host_context(0)
    f, d, b, p, a = (np.ascontiguousarray(g[k], dtype=np.float64) for k in ("freq", "den", "bmag", "bpsi", "alt"))
    n = 2000
    m1 = np.ascontiguousarray(library.smooth_nonuniform_grid(0, 1, n, 10.0))
    m2 = np.ascontiguousarray(np.linspace(0.0, 1.0, n))                     # another grid of the same length
    out = {}
    for tag, m, flag in (("a", m1, _native.FLAG_GRID_STABLE), ("b", m2, _native.FLAG_GRID_STABLE),
                         ("a2", m1, _native.FLAG_GRID_STABLE), ("a_plain", m1, 0), ("b_plain", m2, 0)):
        o = np.empty((64, f.size))
        ctx.set_math(_native.MATH_AUTO)
        _native.raise_for(ctx.vfo_batch(f.ctypes.data, f.size, d.ctypes.data, b.ctypes.data, p.ctypes.data, a.ctypes.data,
                                        64, a.size, a.size, 0, m.ctypes.data, n, _native.MODE_X, o.ctypes.data, flag))
        out[tag] = o
    assert np.array_equal(out["a"], out["a_plain"], equal_nan=True) and np.array_equal(out["a2"], out["a"], equal_nan=True)
    assert np.array_equal(out["b"], out["b_plain"], equal_nan=True)
    assert not np.array_equal(out["a"], out["b"], equal_nan=True)
    assert np.array_equal(out["a"], library.vertical_forward_operator(f, d, b, p, a, "X", n), equal_nan=True)
    one = np.ones(4)
    o1 = np.empty(1)
    rc = ctx.vfo_batch(one.ctypes.data, 1, one.ctypes.data, one.ctypes.data, one.ctypes.data, one.ctypes.data,
                       1, 4, 4, 0, one.ctypes.data, 4, 0, o1.ctypes.data, _native.FLAG_ASYNC)
    assert rc == _native.EINVAL and "device pointers" in _native.last_error()


@pytest.mark.parametrize("slope_deg_per_km,kind", [(0.0005, "linear"), (0.002, "quadratic"), (0.01, "cubic"),
                                                    (0.05, "sin per point")])
def test_every_degree_of_the_sin2_polynomial(lib, slope_deg_per_km, kind):
    """The main loop carries sin^2(psi) inside a segment as a linear, quadratic or cubic polynomial depending on
    how fast the field angle turns (stage_profile), and falls back to sin() per point beyond 3e-4 rad per level:
    each variant - with the top-segment phase at n_points = 20000 and without it at 500 - against the reference
    order (X mode: 1e-9) and the plain-C oracle."""
    from oracle import vfo_c
    g = load_golden("g5_chapman64.npz")
    alt = g["alt"]
    bpsi = 20.0 + np.arange(64)[:, None] + slope_deg_per_km * (alt[None, :] - 80.0)
    freq = g["freq"][::2]
    for n in (500, 20000):
        rows = slice(0, 64) if n == 500 else slice(0, 24)      # both: >= 4096 pairs or a long grid -> main loop
        fast = lib.vertical_forward_operator(freq, g["den"][rows], g["bmag"][rows], bpsi[rows], alt, "X", n)
        slow = lib.vertical_forward_operator(freq, g["den"][rows], g["bmag"][rows], bpsi[rows], alt, "X", n,
                                             math=lib.MATH_FAITHFUL)
        worst = assert_x_mode(fast, slow, tol=1e-9)
        if n == 500:
            vfo_c.require()
            assert_x_mode(fast, vfo_c.virtual_heights_batch(freq, g["den"][rows], g["bmag"][rows], bpsi[rows], alt, "X", n),
                          tol=1e-9)
        print(f"sin^2 {kind} n={n}: fast vs reference order {worst:.2e}")
        vo = lib.vertical_forward_operator(freq, g["den"][rows], g["bmag"][rows], bpsi[rows], alt, "O", n)
        vf = lib.vertical_forward_operator(freq, g["den"][rows], g["bmag"][rows], bpsi[rows], alt, "O", n,
                                           math=lib.MATH_FAITHFUL)
        assert_masks(vo, vf)
        err, ok = rel_err(vo, vf)
        assert err.max() <= 2e-8, (kind, n, err.max())         # default O-mode arithmetic vs reference order everywhere


def test_more_frequencies_than_the_candidate_list_holds(lib):
    """The per-profile candidate list holds 1024 frequencies; a longer sweep takes the path without it (every
    frequency is a work item, the escape test runs per pair) and must give the same values."""
    g = load_golden("g5_chapman64.npz")
    freq = np.linspace(0.3, 17.0, 1100)
    long_ = lib.vertical_forward_operator(freq, g["den"][:8], g["bmag"][:8], g["bpsi"][:8], g["alt"], "X", 300)
    # (1000 frequencies of a 300-point grid go to the X-mode short-grid kernel, 1100 to the general one: another
    #  summation order, ~1e-15 apart; inside the general kernel the list changes nothing, bit for bit)
    head = lib.vertical_forward_operator(freq[:1000], g["den"][:8], g["bmag"][:8], g["bpsi"][:8], g["alt"], "X", 300)
    assert_x_mode(long_[:, :1000], head, tol=1e-12)
    lib.set_option("shortx_kernel", 0)
    try:
        head = lib.vertical_forward_operator(freq[:1000], g["den"][:8], g["bmag"][:8], g["bpsi"][:8], g["alt"], "X", 300)
    finally:
        lib.set_option("shortx_kernel", 1)
    assert np.array_equal(long_[:, :1000], head, equal_nan=True)
    # O mode, n_points = 200: 1000 frequencies go to the short-grid kernel, 1100 (more than its list holds) to the
    # general one - which chooses between its two formulations per wave-iteration, not per point: ~1e-10 apart
    lo = lib.vertical_forward_operator(freq, g["den"][:8], g["bmag"][:8], g["bpsi"][:8], g["alt"], "O", 200)
    ho = lib.vertical_forward_operator(freq[:1000], g["den"][:8], g["bmag"][:8], g["bpsi"][:8], g["alt"], "O", 200)
    assert_masks(lo[:, :1000], ho)
    err, ok = rel_err(lo[:, :1000], ho)
    assert err.max() <= 1e-8, err.max()
    # ... and inside the general kernel the list changes nothing, bit for bit
    lib.set_option("short_kernel", 0)
    try:
        ho_general = lib.vertical_forward_operator(freq[:1000], g["den"][:8], g["bmag"][:8], g["bpsi"][:8], g["alt"], "O", 200)
    finally:
        lib.set_option("short_kernel", 1)
    assert np.array_equal(lo[:, :1000], ho_general, equal_nan=True)
    assert 0.3 < np.isfinite(long_).mean() < 0.8


def test_short_grid_kernels_every_shape(lib):
    """vfo_short_kernel (O mode) and vfo_shortx_kernel (X mode) on grid sizes around their wave-iteration boundaries
    (16 points per pair and wave-iteration; one full trip = 32): X mode against the C oracle; O mode against the
    general kernel (which chooses its arithmetic per wave-iteration, not per point: ~1e-10 apart) and, for three
    sizes, against the NumPy oracle under the parity rule."""
    from oracle import vfo_c, vfo_numpy
    g = load_golden("g5_chapman64.npz")
    args = (g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"])           # 64 x 174 pairs: a long launch
    for n in (2, 3, 15, 16, 17, 31, 32, 33, 48, 64, 65, 129, 200, 511, 1000, 1024):
        got_x = lib.vertical_forward_operator(*args, "X", n)
        vfo_c.require()
        assert_x_mode(got_x, vfo_c.virtual_heights_batch(*args, "X", n), tol=1e-9)
        got_o = lib.vertical_forward_operator(*args, "O", n)
        lib.set_option("short_kernel", 0)
        lib.set_option("shortx_kernel", 0)
        try:
            gen_o = lib.vertical_forward_operator(*args, "O", n)
            gen_x = lib.vertical_forward_operator(*args, "X", n)
        finally:
            lib.set_option("short_kernel", 1)
            lib.set_option("shortx_kernel", 1)
        assert_masks(got_o, gen_o)
        err, ok = rel_err(got_o, gen_o)
        assert err.max(initial=0.0) <= 2e-7, (n, err.max())
        assert_x_mode(got_x, gen_x, tol=1e-9)
        if n in (2, 17, 200):
            with np.errstate(all="ignore"):
                want = vfo_numpy.virtual_heights_batch(*args, "O", n)
            assert_o_mode(got_o, want, oracle_noise(*args, "O", n, runs=12), min_within=0.99)


# ---- fixture G13: profiles taller than LDS holds, NaN-padded densities (made by running the reference) ----------

@pytest.mark.parametrize("trim", [1, 0])
@pytest.mark.parametrize("case,mode,n", [("tall_day", "O", 200), ("tall_day", "X", 2000), ("tall_rag", "O", 200),
                                         ("tall_rag", "X", 500), ("tall_fine", "O", 200), ("tall_fine", "X", 2000)])
def test_profiles_of_more_than_1400_levels_g13(lib, case, mode, n, trim):
    """The reference has no limit on the number of levels (library.py:371-375).  Columns of 3 096 / 2 600 levels
    (uniform / irregular spacing) do not fit LDS, but their bottomsides (peaks at levels 1 290 / 667) do: staged up to
    the highest peak of the launch they stay on the LDS kernels (`trim` 1, the default); with `trim_lds` 0 - and always
    for `tall_fine`, 6 191 levels with the peak at 2 580 - vfo_tall_kernel stages them in global memory."""
    g = load_golden("g13_tall_nanpad.npz")
    a = [g[f"{case}_{k}"] for k in ("freq", "den", "bmag", "bpsi", "alt")]
    assert a[4].size > 1400
    lib.set_option("trim_lds", trim)
    try:
        vh = lib.vertical_forward_operator(*a, mode, n)
        import torch
        t = [torch.as_tensor(np.asarray(x, dtype=np.float64), device="cuda:0") for x in a]
        assert np.array_equal(lib.vertical_forward_operator(*t, mode, n).cpu().numpy(), vh, equal_nan=True)   # (device pre-pass)
    finally:
        lib.set_option("trim_lds", 1)
    want = g[f"{case}_{mode}_{n}_vh"]
    if mode == "X":
        print(case, "X", n, "max rel err", assert_x_mode(vh, want))
    else:
        noise = combined_noise(g[f"{case}_O_{n}_noise"], oracle_noise(*a, "O", n, runs=8, seed=13))
        print(case, "O", n, "max rel err", assert_o_mode(vh, want, noise))


def test_tall_profiles_in_batches_and_work_lists(lib):
    """A batch of tall profiles against the C oracle - 24 profiles x 48 frequencies x 2000 points are cut into 528
    blocks, more than the 512 workgroup slots: a persistent launch, every workgroup re-using its slab; 64 points:
    144 blocks, one slab each - and a mixed work list over them against the single-slice calls."""
    from oracle import vfo_c
    from pyrayhf_amd import synth
    a0, den0, bmag0, bpsi0 = synth.chapman_profiles(24, 1313)
    alt = np.arange(80.0, 700.0, 0.25)                                    # 2 480 levels
    den, bmag, bpsi = (np.array([np.interp(alt, a0, r) for r in x]) for x in (den0, bmag0, bpsi0))
    freq = np.linspace(1.0, 12.0, 48)
    lib.set_option("trim_lds", 0)                 # (the bottomsides of this batch fit LDS: force the slabs here ...)
    try:
        for mode, n in (("X", 2000), ("O", 200), ("X", 64)):
            got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n)
            want = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, mode, n)
            if mode == "X":
                assert_x_mode(got, want)
            else:
                assert_o_mode(got, want, oracle_noise(freq, den, bmag, bpsi, alt, "O", n, runs=6, seed=5))
    finally:
        lib.set_option("trim_lds", 1)
    for mode, n in (("X", 2000), ("O", 200)):     # (... and the trimmed LDS kernels here)
        got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n)
        want = vfo_c.virtual_heights_batch(freq, den, bmag, bpsi, alt, mode, n)
        if mode == "X":
            assert_x_mode(got, want)
        else:
            assert_o_mode(got, want, oracle_noise(freq, den, bmag, bpsi, alt, "O", n, runs=6, seed=5))
    segs = [(0, 10, "O", 200), (10, 24, "X", 2000)]
    mixed = lib.vertical_forward_operator_mixed(freq, den, bmag, bpsi, alt, segs)
    assert np.array_equal(mixed[:10], lib.vertical_forward_operator(freq, den[:10], bmag[:10], bpsi[:10], alt, "O", 200),
                          equal_nan=True)
    assert np.array_equal(mixed[10:], lib.vertical_forward_operator(freq, den[10:], bmag[10:], bpsi[10:], alt, "X", 2000),
                          equal_nan=True)
    # one tall profile, many points: the chunked launch (every workgroup stages the profile into its own slab)
    one = lib.vertical_forward_operator(freq, den[3], bmag[3], bpsi[3], alt, "X", 20000)
    assert_x_mode(one, vfo_c.virtual_heights_batch(freq, den[3:4], bmag[3:4], bpsi[3:4], alt, "X", 20000)[0])
    # a profile of 1 401 levels is tall, one of 1 400 is not: the same inputs cut at either size agree where both see
    # the whole bottomside
    k = int(np.argmax(den[5])) + 2
    assert k < 1400
    lo = lib.vertical_forward_operator(freq, den[5, :1400], bmag[5, :1400], bpsi[5, :1400], alt[:1400], "X", 2000)
    hi = lib.vertical_forward_operator(freq, den[5, :1401], bmag[5, :1401], bpsi[5, :1401], alt[:1401], "X", 2000)
    assert_x_mode(hi, lo, tol=1e-9)


@pytest.mark.parametrize("first", [300, 200])
def test_density_padded_with_nan_is_cut_at_the_padding_g13(lib, first):
    """np.argmax returns the first NaN (library.py:371): a density column padded with NaN from level `first` up is
    evaluated as the reference evaluates it - levels [0, first)."""
    g = load_golden("g13_tall_nanpad.npz")
    a = [g["nanpad_freq"], g[f"nanpad_{first}_den"], g["nanpad_bmag"], g["nanpad_bpsi"], g["nanpad_alt"]]
    vx = lib.vertical_forward_operator(*a, "X", 200)
    assert_x_mode(vx, g[f"nanpad_{first}_X_200_vh"])
    vo = lib.vertical_forward_operator(*a, "O", 200)
    noise = combined_noise(g[f"nanpad_{first}_O_200_noise"], oracle_noise(*a, "O", 200, runs=8, seed=first))
    assert_o_mode(vo, g[f"nanpad_{first}_O_200_vh"], noise)
    # the same on GPU-resident inputs
    import torch
    t = [torch.as_tensor(np.asarray(x, dtype=np.float64), device="cuda:0") for x in a]
    assert np.array_equal(lib.vertical_forward_operator(*t, "X", 200).cpu().numpy(), vx, equal_nan=True)


def _nanfield_inputs(g, case):
    day = load_golden("g4_day_night.npz")
    a = {"freq": g["nanpad_freq"], "den": day["Day_den"].copy(), "bmag": day["Day_bmag"].copy(),
         "bpsi": day["Day_bpsi"].copy(), "alt": day["Day_alt"].copy()}
    a[str(g[f"nanfield_{case}_col"])][g[f"nanfield_{case}_levels"]] = np.nan
    return [a[k] for k in ("freq", "den", "bmag", "bpsi", "alt")]


@pytest.mark.parametrize("case", ["alt_100", "alt_500", "bmag_1", "bmag_100", "bmag_400", "bpsi_1", "bpsi_100",
                                  "bpsi_257", "bpsi_50_200"])
def test_nan_in_altitude_field_strength_or_angle_behaves_as_in_the_reference_g13(lib, case):
    """A NaN altitude: the whole trace is NaN (np.min(alt), library.py:507).  A NaN |B| below the peak: the X-mode
    trace is NaN (running maximum of X + Y, :389); in O mode - and a NaN psi in either mode - the grid points of the
    two segments next to the level are blanked by np.interp and the sum skips them (:288), except a point that sits
    ON the level below (grid point 0 on level 0: `bpsi_1`, `bmag_1`).  Above the peak nothing is read (`bmag_400`)."""
    g = load_golden("g13_tall_nanpad.npz")
    a = _nanfield_inputs(g, case)
    for mode in "OX":
        for n in (200, 2000):
            want = g[f"nanfield_{case}_{mode}_{n}_vh"]
            got = lib.vertical_forward_operator(*a, mode, n)
            assert_masks(got, want)
            if not np.isfinite(want).any():
                continue
            if mode == "X":
                assert_x_mode(got, want)
            else:
                assert_o_mode(got, want, oracle_noise(*a, "O", n, runs=8, seed=n))


def test_inputs_written_through_the_bar_are_never_stale(lib):
    """Small host-buffer calls write their inputs straight into device memory from the CPU (large BAR, option
    direct_upload).  Nothing orders those stores but the launch that follows them, and the GPU's caches are not
    snooped: alternate between profiles, frequencies sweeps and grid sizes a few hundred times and require every
    result to be bit-identical to what the staged upload (direct_upload = 0) gives for the same inputs."""
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(6, 4242)
    sweeps = [synth.sounder_frequencies(1), np.linspace(1.0, 9.0, 40), np.linspace(2.0, 12.0, 174)]
    cases = [(p, k, mode, n) for p in range(6) for k in range(3) for mode, n in (("O", 200), ("X", 200), ("X", 2000))]
    lib.set_option("direct_upload", 0)
    try:
        want = [lib.vertical_forward_operator(sweeps[k], den[p], bmag[p], bpsi[p], alt, mode, n) for p, k, mode, n in cases]
    finally:
        lib.set_option("direct_upload", 1)
    rng = np.random.default_rng(7)
    for it in range(400):
        i = int(rng.integers(len(cases)))
        p, k, mode, n = cases[i]
        got = lib.vertical_forward_operator(sweeps[k], den[p], bmag[p], bpsi[p], alt, mode, n)
        assert np.array_equal(got, want[i], equal_nan=True), (it, cases[i])
    # a 2-row batch and a call whose inputs exceed the direct limit in between
    big = lib.vertical_forward_operator(sweeps[2], den, bmag, bpsi, alt, "X", 200)
    for i in (0, 17, 33):
        p, k, mode, n = cases[i]
        assert np.array_equal(lib.vertical_forward_operator(sweeps[k], den[p], bmag[p], bpsi[p], alt, mode, n), want[i], equal_nan=True)
    assert np.array_equal(big[2], lib.vertical_forward_operator(sweeps[2], den[2], bmag[2], bpsi[2], alt, "X", 200), equal_nan=True)


def test_field_strength_that_is_nan_everywhere_gives_nan_as_in_the_reference(lib):
    """np.nanmax of an all-NaN |Y| array is NaN, which is not below the isotropic tolerance (library.py:201): the
    magnetised formulas run on NaN and every virtual height is NaN - not the isotropic answer."""
    from oracle import vfo_numpy as orc
    g = load_golden("g4_day_night.npz")
    b = np.full_like(g["Day_bmag"], np.nan)
    for mode in "OX":
        with np.errstate(all="ignore"):
            want = orc.virtual_heights(g["freq"], g["Day_den"], b, g["Day_bpsi"], g["Day_alt"], mode, 200)
        got = lib.vertical_forward_operator(g["freq"], g["Day_den"], b, g["Day_bpsi"], g["Day_alt"], mode, 200)
        assert not np.isfinite(want).any() and np.array_equal(np.isnan(got), np.isnan(want)), mode
