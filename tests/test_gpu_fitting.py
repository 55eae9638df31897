"""Batched residual / brute-force driver against rows produced by the reference's residual_VH
itself (fixture G11, library.py:595-669), against the oracle, and a self-consistency fit in the spirit of the
reference's test_zero_residual_when_parameters_match (test_core.py:279-320)."""

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_residual_rows_match_the_oracle():
    from oracle import vfo_numpy as orc
    from pyrayhf_amd import fitting
    g = load_golden("g5_chapman64.npz")
    freq = g["freq"][5:140:3]
    obs = g["X_2000_vh"][3, 5:140:3].copy()
    obs[np.isnan(obs)] = 250.0
    res, cost = fitting.residual_VH_batch(freq, obs, g["den"][:16], g["bmag"][3], g["bpsi"][3], g["alt"], "X", 2000)
    model = orc.virtual_heights_batch(freq, g["den"][:16], np.tile(g["bmag"][3], (16, 1)),
                                      np.tile(g["bpsi"][3], (16, 1)), g["alt"], "X", 2000)
    want = orc.residual_rows(obs, model)
    assert res.shape == want.shape == (16, freq.size)
    np.testing.assert_allclose(res, want, rtol=0, atol=1e-6)
    np.testing.assert_allclose(cost, (want ** 2).sum(axis=1), rtol=1e-9)
    # the candidate that generated the observations has a ~zero residual where both reflect
    both = np.isfinite(model[3])
    assert np.max(np.abs(res[3][both])) < 1e-6


@pytest.mark.parametrize("mode", ["O", "X"])
def test_residual_rows_match_the_reference_g11(mode):
    """prhf_vfo_residual_f64 against the reference's own residual_VH output: EDP -> operator -> NaN fill
    (max(nanmean|vh|, 100), library.py:664-665) -> vh_obs - vh_model (library.py:668)."""
    from pyrayhf_amd import fitting
    g = load_golden("g11_residual.npz")
    for name in g["cases"]:
        want = g[f"{name}_{mode}_residual"]
        obs = g[f"{name}_{mode}_vh_obs"]
        res, cost, vh = fitting.residual_VH_batch(g[f"{name}_freq"], obs, g[f"{name}_edp"], g[f"{name}_bmag"],
                                                  g[f"{name}_bpsi"], g[f"{name}_alt"], mode,
                                                  int(g[f"{name}_{mode}_n_points"]), return_vh=True)
        assert res.shape == want.shape
        assert np.array_equal(np.isnan(res), np.isnan(want)), (name, mode)
        ok = np.isfinite(want)
        if name == "all_nan":
            assert not ok.any() and np.isnan(cost).all() and np.isnan(vh).all()
            continue
        # residuals are differences of heights: compare on the scale of the heights (km)
        scale = np.abs(np.broadcast_to(obs, want.shape)[ok]) + np.abs(want[ok])
        err = np.abs(res[ok] - want[ok]) / scale
        if mode == "X":
            assert err.max() <= 1e-8, (name, err.max())
        else:
            # O mode: the modeled traces under the per-pair rule of tests/parity.py, with the floors that
            # oracle/gen_golden.py recorded from the REFERENCE for every candidate EDP (+-1 ulp on its inputs, 24
            # runs) and the rounding noise of the same rows; the residual rows are those traces subtracted from vh_obs
            from parity import assert_o_mode, combined_noise
            floor = combined_noise(g[f"{name}_O_noise"], g[f"{name}_O_noise_rounding"])
            assert_o_mode(vh, g[f"{name}_O_vh_model"], floor)
            height = np.abs(g[f"{name}_O_vh_model"])
            limit = np.minimum(1e-3, np.maximum(1e-6, 4.0 * floor))
            both = ok & np.isfinite(g[f"{name}_O_vh_model"])
            assert np.all(np.abs(res[both] - want[both]) <= limit[both] * height[both]), name
        filled = np.isnan(vh) & ok
        if name == "low_layer":
            assert filled.any()
            np.testing.assert_array_equal(res[filled], np.broadcast_to(obs, want.shape)[filled] - 100.0)
        np.testing.assert_allclose(cost[np.isfinite(cost)], np.nansum(want ** 2, axis=1)[np.isfinite(cost)],
                                   rtol=1e-3 if mode == "O" else 1e-7, atol=1e-6)   # the generating node: ~0


def test_all_nan_candidate_gives_nan_cost_and_is_skipped():
    from pyrayhf_amd import fitting
    g = load_golden("g5_chapman64.npz")
    freq = np.array([25.0, 30.0])                       # above every foF2: every candidate escapes
    res, cost = fitting.residual_VH_batch(freq, np.array([300.0, 320.0]), g["den"][:4], g["bmag"][0], g["bpsi"][0],
                                          g["alt"], "O", 100)
    assert np.all(np.isnan(res)) and np.all(np.isnan(cost))          # np.maximum(nan, 100) = nan, library.py:664
    with pytest.raises(ValueError):
        fitting.brute_force_fit(freq, np.array([300.0, 320.0]), g["den"][:4], g["bmag"][0], g["bpsi"][0], g["alt"],
                                "O", 100)


def test_brute_force_recovers_the_generating_profile():
    """Synthetic truth: a Chapman F2 layer; candidates scan hmF2 and the scale height around it."""
    from pyrayhf_amd import fitting
    alt = np.arange(80.0, 700.0, 1.0)

    def layer(hm, h):
        z = (alt - hm) / h
        return 8e11 * np.exp(0.5 * (1 - z - np.exp(-z))) + 1e11 * np.exp(0.5 * (1 - (alt - 110) / 8 - np.exp(-(alt - 110) / 8)))

    bmag = 4.5e-5 * ((6371.0 + 80.0) / (6371.0 + alt)) ** 3
    bpsi = np.full(alt.size, 35.0)
    hms, hs = np.arange(280.0, 321.0, 2.0), np.arange(40.0, 61.0, 2.0)
    grid = [(hm, h) for hm in hms for h in hs]
    den = np.array([layer(hm, h) for hm, h in grid])
    truth = grid.index((300.0, 50.0))
    freq = np.arange(2.0, 8.0, 0.25)
    from pyrayhf_amd import library
    obs = library.vertical_forward_operator(freq, den[truth], bmag, bpsi, alt, "O", 200)
    obs_noisy = obs.copy()
    obs_noisy[3] = np.nan                                # a missing sounding is filtered out (library.py:742)
    best, cost, vh_best, f_used = fitting.brute_force_fit(freq[::-1], obs_noisy[::-1], den, bmag, bpsi, alt, "O", 200)
    # (the lone profile behind `obs` runs in the general kernel, the 231 candidates in the short-grid kernel: the two
    #  choose between the reduced algebra and the reference's order per wave-iteration and per point - ~1e-10 apart)
    assert best == truth and cost[truth] < 1e-10 and f_used.size == freq.size - 1
    np.testing.assert_allclose(vh_best, np.delete(obs, 3), rtol=1e-9)       # the modeled trace itself, NaNs kept
    assert np.all(np.diff(f_used) > 0)
    assert cost.shape == (len(grid),) and np.sum(cost < 1.0) == 1


def test_peak_density_from_trace_formulae():
    from pyrayhf_amd import fitting, library
    assert fitting.peak_density_from_trace(9.0) == library.freq2den(9.0e6) * 1.0001          # library.py:768
    alt = np.arange(80.0, 700.0, 1.0)
    bmag = np.full(alt.size, 4e-5)
    fc = 4e-5 * 2.799249247e10
    want = library.freq2den(np.sqrt(9.0e6 ** 2 - 9.0e6 * fc)) * 1.0001                       # library.py:774-778
    assert fitting.peak_density_from_trace(9.0, "X", alt=alt, bmag=bmag, hmf2=300.0) == want


def test_shared_field_rows_equal_their_broadcast():
    """PRHF_FLAG_SHARED_FIELD: one bmag / bpsi row for every candidate (what a fit has) gives, bit for bit, what the
    (P, N_alt) copies of that row give - through the residual entry point and through the operator."""
    import time
    from pyrayhf_amd import fitting, library, synth
    alt, den, bmag, bpsi = synth.chapman_profiles(3000, 31)
    freq = np.arange(1.0, 12.0, 0.1)
    obs = library.vertical_forward_operator(freq, den[7], bmag[0], bpsi[0], alt, "O", 200)
    keep = np.isfinite(obs)
    t0 = time.perf_counter()
    r1, c1, v1 = fitting.residual_VH_batch(freq[keep], obs[keep], den, bmag[0], bpsi[0], alt, "O", 200, return_vh=True)
    t1 = time.perf_counter()
    wide_b, wide_p = np.tile(bmag[0], (3000, 1)), np.tile(bpsi[0], (3000, 1))
    r2, c2, v2 = fitting.residual_VH_batch(freq[keep], obs[keep], den, wide_b, wide_p, alt, "O", 200, return_vh=True)
    t2 = time.perf_counter()
    assert np.array_equal(r1, r2, equal_nan=True) and np.array_equal(c1, c2, equal_nan=True) and np.array_equal(v1, v2, equal_nan=True)
    assert int(np.nanargmin(c1)) == 7 and c1[7] < 1e-10          # (a lone profile takes another kernel: ~1e-10 apart)
    print(f"3000 candidates x {keep.sum()} freqs from host arrays: shared field rows {1e3 * (t1 - t0):.2f} ms, broadcast copies {1e3 * (t2 - t1):.2f} ms")
    a = library.vertical_forward_operator(freq, den[:64], bmag[0], bpsi[0], alt, "X", 2000)
    b = library.vertical_forward_operator(freq, den[:64], wide_b[:64], wide_p[:64], alt, "X", 2000)
    assert a.shape == (64, freq.size) and np.array_equal(a, b, equal_nan=True)
    import torch
    dev = torch.device("cuda:0")
    t = [torch.as_tensor(x, device=dev) for x in (freq, den[:64], bmag[0], bpsi[0], alt)]
    c = library.vertical_forward_operator(*t, "X", 2000)
    assert np.array_equal(c.cpu().numpy(), a, equal_nan=True)


def _chapman_builder(F2, F1, E, alt, bottom_type):
    """A stand-in for the reference's PyIRI EDP builders (library.py:557-586): alpha-Chapman F2 + E layers from the
    dictionaries minimize_parameters hands over (the shape of oracle/gen_golden.py's G11 stand-in)."""
    thick = F2['B_bot'] if bottom_type == 'B_bot' else F2['B0']
    z = (alt - F2['hm'].ravel()[0]) / thick.ravel()[0]
    ze = (alt - E['hm'].ravel()[0]) / E['B_bot'].ravel()[0]
    return (F2['Nm'].ravel()[0] * np.exp(0.5 * (1.0 - z - np.exp(-z)))
            + E['Nm'].ravel()[0] * np.exp(0.5 * (1.0 - ze - np.exp(-ze))))


def _layer_dicts(nm, hm, bb):
    one = lambda v: np.array([[[v]]])                                           # noqa: E731
    return ({"Nm": one(nm), "hm": one(hm), "B_bot": one(bb)}, {"Nm": one(0.0), "hm": one(200.0), "B_bot": one(30.0)},
            {"Nm": one(3e10), "hm": one(110.0), "B_bot": one(8.0)})


def test_brute_grid_is_the_end_exclusive_arange():
    from pyrayhf_amd import fitting
    nodes = fitting.brute_grid(np.array([[[300.0]]]), 20.0, 1.0)
    assert nodes[0] == 240.0 and nodes[-1] == 359.0 and nodes.size == 120          # np.mgrid[slice(240, 360, 1)]
    assert fitting.brute_grid(45.0, 20.0, 2.0).tolist() == [36.0, 38.0, 40.0, 42.0, 44.0, 46.0, 48.0, 50.0, 52.0]


@pytest.mark.parametrize("mode,n_points", [("X", 200), ("O", 200)])
def test_minimize_parameters_recovers_the_generating_layer(mode, n_points):
    """The reference's minimize_parameters signature and return triple (library.py:672-674, :821-825) over a 3450-node
    (hmF2 x B_bot) grid in ONE launch; X mode also node for node against a Python loop over the oracle."""
    from oracle import vfo_numpy as orc
    from pyrayhf_amd import fitting, library
    alt = np.arange(80.0, 500.0, 1.0)
    b_mag = 4.6e-5 * ((6371.0 + 80.0) / (6371.0 + alt)) ** 3
    b_psi = 35.0 + 0.002 * (alt - alt[0])
    f_in0 = np.arange(1.5, 9.6, 0.25)
    # the fit takes NmF2 from the highest sounding (library.py:760-778; X mode: with |B| at the background hmF2): the
    # generating layer has exactly that peak density, and its (hmF2, B_bot) = (300.5, 46) is a node of the search grid
    # (the peak half way between two levels: the level below it - the top of the bottomside, library.py:371-375 - is
    #  within the 0.01 % by which the fit raises NmF2, so the highest sounding still reflects)
    nm_true = fitting.peak_density_from_trace(f_in0[-1], mode, alt=alt, bmag=b_mag, hmf2=322.0)
    F2_true, F1, E = _layer_dicts(nm_true, 300.5, 46.0)
    truth = _chapman_builder(F2_true, F1, E, alt, 'B_bot')
    vh_obs0 = library.vertical_forward_operator(f_in0, truth, b_mag, b_psi, alt, mode, n_points)
    assert np.isfinite(vh_obs0).all()
    vh_obs0[4] = np.nan                                    # a missing sounding (filtered, library.py:741-742)
    F2_start, _, _ = _layer_dicts(1.0e12, 322.0, 40.0)     # the background the search is centred on
    vh, edp, F2_fit = fitting.minimize_parameters(F2_start, F1, E, f_in0[::-1].copy(), vh_obs0[::-1].copy(), alt, b_mag, b_psi,
                                                  'brute', 25.0, 1.0, mode, n_points, 'B_bot', edp_builder=_chapman_builder)
    hm_nodes, bb_nodes = fitting.brute_grid(322.0, 25.0, 1.0), fitting.brute_grid(40.0, 25.0, 1.0)
    assert hm_nodes.size * bb_nodes.size >= 3000
    assert F2_fit['hm'].shape == F2_start['Nm'].shape and float(F2_fit['hm'].squeeze()) == 300.5
    assert float(F2_fit['B_bot'].squeeze()) == 46.0
    keep = np.isfinite(f_in0 + vh_obs0)
    f_max = np.sort(f_in0[keep])[-1]
    assert float(F2_fit['Nm'].squeeze()) == fitting.peak_density_from_trace(f_max, mode, alt=alt, bmag=b_mag, hmf2=322.0)
    assert vh.shape == f_in0.shape and edp.shape == alt.shape
    both = np.isfinite(vh[::-1]) & np.isfinite(vh_obs0)
    assert both.sum() >= 30 and np.max(np.abs(vh[::-1][both] - vh_obs0[both])) < 1e-6      # the final trace (:821-824)
    assert float(F2_start['hm'].squeeze()) == 322.0                              # the inputs are not mutated
    if mode == "X":
        # node for node: the batched costs against a Python loop of oracle evaluations + the restated residual_VH
        f_s, obs_s = fitting._sorted_finite(f_in0, vh_obs0)
        nm = float(F2_fit['Nm'].squeeze())
        nodes = [(hm, bb) for hm in hm_nodes for bb in bb_nodes]
        pick = list(range(0, len(nodes), 7))                                     # every 7th node: ~500 oracle calls
        den = np.array([_chapman_builder(_layer_dicts(nm, *nodes[k])[0], F1, E, alt, 'B_bot') for k in pick])
        model = np.array([orc.virtual_heights(f_s, d, b_mag, b_psi, alt, mode, n_points) for d in den])
        want = orc.residual_rows(obs_s, model)
        res, cost = fitting.residual_VH_batch(f_s, obs_s, den, b_mag, b_psi, alt, mode, n_points)
        assert np.array_equal(np.isnan(res), np.isnan(want))
        np.testing.assert_allclose(res, want, rtol=0, atol=1e-5)                 # km; 1e-8 of the heights
        np.testing.assert_allclose(cost, (want ** 2).sum(axis=1), rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError, match="B0 and B1 are not provided"):
        fitting.minimize_parameters(F2_start, F1, E, f_in0, vh_obs0, alt, b_mag, b_psi, bottom_type='B0_B1',
                                    edp_builder=_chapman_builder)
    with pytest.raises(NotImplementedError):              # lmfit's sampler is not restated (fitting.resolve_method)
        fitting.minimize_parameters(F2_start, F1, E, f_in0, vh_obs0, alt, b_mag, b_psi, 'emcee', edp_builder=_chapman_builder)


@pytest.mark.parametrize("method,tol_km", [("leastsq", None), ("least_squares", None), ("powell", 1e-2),
                                           ("levenberg-marquardt", None), ("lbfgsb", None),
                                           ("differential_evolution", 1e-2)])
def test_minimize_parameters_other_lmfit_methods(method, tol_km):
    """The reference forwards `method` to lmfit.minimize (library.py:794-798): the local and population optimisers lmfit
    runs for those names (restated, fitting.resolve_method), every residual evaluated by the fused kernel - the
    forward-difference Jacobian of Levenberg-Marquardt and a generation of differential evolution as one launch each.
    The cost surface is the reference's: the bottomside ends at the level below the density peak (library.py:371-375),
    so the trace near foF2 jumps whenever hmF2 crosses half a level, and a gradient search stops at the first such
    step (which is why the reference's default is the grid search).  Asked of every method: a lower cost than the
    start, inside the bounds; of Powell's line searches and of differential evolution: the generating layer."""
    from pyrayhf_amd import fitting, library
    alt = np.arange(80.0, 500.0, 1.0)
    b_mag = 4.6e-5 * ((6371.0 + 80.0) / (6371.0 + alt)) ** 3
    b_psi = 35.0 + 0.002 * (alt - alt[0])
    f_in0 = np.arange(1.5, 9.6, 0.25)
    nm_true = fitting.peak_density_from_trace(f_in0[-1], "X", alt=alt, bmag=b_mag, hmf2=304.0)
    F2_true, F1, E = _layer_dicts(nm_true, 300.5, 46.0)
    truth = _chapman_builder(F2_true, F1, E, alt, 'B_bot')
    vh_obs0 = library.vertical_forward_operator(f_in0, truth, b_mag, b_psi, alt, "X", 200)
    assert np.isfinite(vh_obs0).all()
    F2_start, _, _ = _layer_dicts(1.0e12, 304.0, 44.0)
    calls = []
    real = fitting.residual_VH_batch

    def counting(*a, **k):
        calls.append(np.atleast_2d(a[2]).shape[0])
        return real(*a, **k)
    fitting.residual_VH_batch = counting
    try:
        vh, edp, F2_fit = fitting.minimize_parameters(F2_start, F1, E, f_in0, vh_obs0, alt, b_mag, b_psi, method, 20.0,
                                                      1.0, "X", 200, 'B_bot', edp_builder=_chapman_builder)
    finally:
        fitting.residual_VH_batch = real
    hm, bb = float(F2_fit['hm'].squeeze()), float(F2_fit['B_bot'].squeeze())
    assert 304.0 * 0.8 <= hm <= 304.0 * 1.2 and 44.0 * 0.8 <= bb <= 44.0 * 1.2
    assert float(F2_fit['Nm'].squeeze()) == nm_true
    start_edp = _chapman_builder(_layer_dicts(nm_true, 304.0, 44.0)[0], F1, E, alt, 'B_bot')
    _, cost = fitting.residual_VH_batch(f_in0, vh_obs0, np.stack([start_edp, edp]), b_mag, b_psi, alt, "X", 200)
    assert cost[1] < cost[0], (method, hm, bb, cost)
    if tol_km is not None:
        assert abs(hm - 300.5) < tol_km and abs(bb - 46.0) < tol_km, (method, hm, bb)
        assert np.nanmax(np.abs(vh - vh_obs0)) < 1.0
    assert F2_fit['hm'].shape == F2_start['Nm'].shape and vh.shape == f_in0.shape and edp.shape == alt.shape
    assert float(F2_start['hm'].squeeze()) == 304.0
    if method == "leastsq":
        assert 3 in calls                                  # f(x) and the two forward steps of the Jacobian: one launch
    if method == "differential_evolution":
        assert max(calls) >= 30                            # a generation (15 x 2 members) at a time
    print(f"{method}: hmF2 {hm:.6f}, B_bot {bb:.6f}, cost {cost[0]:.4g} -> {cost[1]:.4g}, {len(calls)} launches, "
          f"{sum(calls)} residual rows")


def test_residual_batch_on_gpu_resident_candidates():
    """Candidates that live on the GPU (torch) give the rows of the same candidates passed as NumPy arrays, bit for bit."""
    import torch
    from pyrayhf_amd import fitting, synth
    alt, den, bmag, bpsi = synth.chapman_profiles(300, 99)
    freq = np.arange(1.0, 11.0, 0.2)
    obs = np.full(freq.size, 250.0)
    want = fitting.residual_VH_batch(freq, obs, den, bmag[0], bpsi[0], alt, "O", 200, return_vh=True)
    dev = torch.device("cuda:0")
    got = fitting.residual_VH_batch(freq, obs, torch.as_tensor(den, device=dev), torch.as_tensor(bmag[0], device=dev),
                                    bpsi[0], alt, "O", 200, return_vh=True)
    for a, b in zip(got, want):
        assert a.is_cuda and np.array_equal(a.cpu().numpy(), b, equal_nan=True)


def test_model_VH_and_residual_VH_keep_the_reference_call():
    """model_VH (library.py:512-592) and residual_VH (:595-669) with the reference's positional arguments; the profile
    builder passed as a keyword (PyIRI, their default, is absent here).  residual_VH is the row residual_VH_batch
    gives for the same candidate - the rows fixture G11 pins to the reference's own residual_VH - bit for bit, with
    plain numbers or lmfit-style objects as parameters, and the layer dictionaries are left as they were."""
    from pyrayhf_amd import fitting, library
    alt = np.arange(80.0, 500.0, 1.0)
    b_mag = 4.6e-5 * ((6371.0 + 80.0) / (6371.0 + alt)) ** 3
    b_psi = 35.0 + 0.002 * (alt - alt[0])
    f_in = np.arange(1.5, 12.0, 0.25)
    F2, F1, E = _layer_dicts(6.0e11, 300.5, 46.0)
    vh, edp = fitting.model_VH(F2, F1, E, f_in, alt, b_mag, b_psi, 'X', 200, 'B_bot', edp_builder=_chapman_builder)
    assert np.array_equal(edp, _chapman_builder(F2, F1, E, alt, 'B_bot'))
    assert np.array_equal(vh, library.vertical_forward_operator(f_in, edp, b_mag, b_psi, alt, 'X', 200), equal_nan=True)
    assert np.isnan(vh).any() and np.isfinite(vh).any()
    obs = np.where(np.isfinite(vh), vh + 1.0, 260.0)

    class P:                                   # what lmfit.Parameters hands over: objects with a .value
        def __init__(self, v):
            self.value = v
    for params in ({"NmF2": 5.0e11, "hmF2": 290.0, "B_bot": 40.0}, {"NmF2": P(5.0e11), "hmF2": P(290.0), "B_bot": P(40.0)}):
        res = fitting.residual_VH(params, F2, F1, E, f_in, obs, alt, b_mag, b_psi, 'X', 200, 'B_bot',
                                  edp_builder=_chapman_builder)
        cand = _chapman_builder(_layer_dicts(5.0e11, 290.0, 40.0)[0], F1, E, alt, 'B_bot')
        want = fitting.residual_VH_batch(f_in, obs, cand[None, :], b_mag, b_psi, alt, 'X', 200, return_cost=False)[0]
        assert res.shape == f_in.shape and np.array_equal(res, want)
        model = library.vertical_forward_operator(f_in, cand, b_mag, b_psi, alt, 'X', 200)
        filled = np.isnan(model)
        assert filled.any() and np.all(res[filled] == obs[filled] - max(np.nanmean(np.abs(model)), 100.0))   # :664-665
    assert float(F2['hm'].squeeze()) == 300.5 and float(F2['Nm'].squeeze()) == 6.0e11
