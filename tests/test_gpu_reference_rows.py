"""BASELINE configs 4 and 5 pinned to the reference at their own shapes (fixtures G14 / G15, made by running
`library.py:459-509` on the first rows of the seeded batches, oracle/gen_golden.py): the HIP path must reproduce
those rows three ways - evaluated alone, inside rank 0's per-GPU shard (12 500 rows; the cut of the mixed list), and
(tests/test_gpu_full_size.py) inside the full launch, which is bit-identical to the shards.

X mode: 1e-8 per pair (BASELINE asks 1e-4; the reference's own +-1 ulp response at these shapes is 4e-11).
O mode: the parity rule of tests/parity.py with the floors recorded from the reference (+ the oracle-made rounding
noise, stored beside them and labelled)."""

import numpy as np
import pytest

from conftest import load_golden
from parity import assert_o_mode, assert_o_mode_reference_noise_alone, assert_x_mode, combined_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pyrayhf_amd import library
    return library


def test_config4_rows_alone_and_inside_the_shard_g14(lib):
    from pyrayhf_amd import dist as pdist, synth
    g = load_golden("g14_config4_rows.npz")
    want = g["X_20000_vh"]
    alone = lib.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "X", 20000)
    worst = assert_x_mode(alone, want)
    # one row by itself: the chunked single-profile launch (another summation order)
    one = lib.vertical_forward_operator(g["freq"], g["den"][3], g["bmag"][3], g["bpsi"][3], g["alt"], "X", 20000)
    assert_x_mode(one, want[3])
    # rank 0's shard of the 8-rank cut, as bench.py and the N = 8 run launch it
    lo, hi = pdist.shard_bounds(100000, 8, 0)
    alt, den, bmag, bpsi = synth.chapman_profiles(100000, 20260004, rows=slice(lo, hi))
    assert np.array_equal(den[:16], g["den"])
    shard = lib.vertical_forward_operator(g["freq"], den, bmag, bpsi, alt, "X", 20000)
    assert shard.shape == (12500, 256)
    assert_x_mode(shard[:16], want)
    assert np.array_equal(shard[:16], alone, equal_nan=True)       # (a launch's rows do not depend on their neighbours)
    print(f"G14: 16 x 256 X/20000 against the reference: worst {worst:.2e} (reference noise {np.nanmax(g['X_20000_noise']):.1e})")


def _check_slice(got, g, mode, n, label):
    want = g[f"{mode}_{n}_vh"]
    if mode == "X":
        worst = assert_x_mode(got, want)
    else:
        worst = assert_o_mode(got, want, combined_noise(g[f"O_{n}_noise"], g[f"O_{n}_noise_rounding"]))
    print(f"G15 {label} {mode}/{n}: worst {worst:.2e}")


def test_config5_rows_alone_inside_the_shard_and_inside_a_mixed_list_g15(lib):
    from bench import CONFIG5_SEGMENTS
    from pyrayhf_amd import dist as pdist, synth
    g = load_golden("g15_config5_rows.npz")
    freq, alt = g["freq"], g["alt"]
    slices = [(int(a), int(b), "OX"[int(m)], int(n)) for a, b, m, n in g["slices"]]
    assert slices == [tuple(s) for s in CONFIG5_SEGMENTS]
    # (1) every slice's rows alone, one homogeneous launch each
    for _p0, _p1, mode, n in slices:
        got = lib.vertical_forward_operator(freq, g[f"{mode}_{n}_den"], g[f"{mode}_{n}_bmag"], g[f"{mode}_{n}_bpsi"], alt, mode, n)
        _check_slice(got, g, mode, n, "alone")
    # (2) the 32 rows as ONE small mixed work list
    den = np.concatenate([g[f"{m}_{n}_den"] for _a, _b, m, n in slices])
    bmag = np.concatenate([g[f"{m}_{n}_bmag"] for _a, _b, m, n in slices])
    bpsi = np.concatenate([g[f"{m}_{n}_bpsi"] for _a, _b, m, n in slices])
    segs = [(8 * i, 8 * i + 8, m, n) for i, (_a, _b, m, n) in enumerate(slices)]
    mixed = lib.vertical_forward_operator_mixed(freq, den, bmag, bpsi, alt, segs)
    for i, (_a, _b, mode, n) in enumerate(slices):
        _check_slice(mixed[8 * i: 8 * i + 8], g, mode, n, "small list")
    # (3) rank 0's cut of the real list (6 250 rows, the bench's config5_shard leg): the fixture rows are the first
    # eight of every slice there
    rows, local = pdist.shard_segments(CONFIG5_SEGMENTS, 8, 0)
    a5, den5, bmag5, bpsi5 = synth.chapman_profiles(50000, 20260005, rows=rows)
    shard = lib.vertical_forward_operator_mixed(freq, den5, bmag5, bpsi5, a5, local)
    assert shard.shape == (6250, 512)
    for (l0, _l1, mode, n) in local:
        assert np.array_equal(rows[l0: l0 + 8], g[f"{mode}_{n}_rows"])
        _check_slice(shard[l0: l0 + 8], g, mode, n, "shard")


# pairs that may lie beyond max(1e-6, 4 x the reference's recorded noise) per O-mode slice of G15 - the ones only the
# rounding noise (NumPy's one-ulp pow) explains; measured on the GPU and fixed here so that a regression shows up
G15_BEYOND_REFERENCE_NOISE = {200: 0, 2000: 1}       # (O/2000: one pair at 2.2e-6, inside 4 x its rounding-noise floor)


@pytest.mark.parametrize("n_points", [200, 2000])
def test_config5_o_rows_against_the_reference_noise_alone_g15(lib, n_points):
    """G15's O/200 and O/2000 rows (8 rows x 512 frequencies of config 5 each) under SURVEY 8(d)'s rule as written:
    the limit comes from the reference's recorded `*_noise` alone, the exceptions are counted and each stays below
    1e-5.  Default and reference-order arithmetic."""
    g = load_golden("g15_config5_rows.npz")
    for math in (None, lib.MATH_FAITHFUL):
        got = lib.vertical_forward_operator(g["freq"], g[f"O_{n_points}_den"], g[f"O_{n_points}_bmag"],
                                            g[f"O_{n_points}_bpsi"], g["alt"], "O", n_points, math=math)
        count, worst = assert_o_mode_reference_noise_alone(got, g[f"O_{n_points}_vh"], g[f"O_{n_points}_noise"],
                                                           allowed_beyond=G15_BEYOND_REFERENCE_NOISE[n_points],
                                                           min_within=0.99)
        print(f"G15 O/{n_points} math={math}: {count} pairs beyond the reference-noise rule, worst of them {worst:.2e}")
