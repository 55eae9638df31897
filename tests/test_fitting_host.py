"""Host logic of the fitting driver's optimisers (no GPU): how a `method` name is resolved (what the reference's
lmfit.minimize call, library.py:794-798, runs for it), the bounds transform, and each optimiser family on a synthetic
residual function - the drivers only ever see `residuals(nodes (P, 2)) -> (P, F)`."""

import numpy as np
import pytest

from pyrayhf_amd import fitting


def test_method_names_resolve_as_lmfit_resolves_them():
    r = fitting.resolve_method
    assert r('brute') == ('brute', None)
    assert r('leastsq') == ('leastsq', None) and r('LeastSq') == ('leastsq', None)
    assert r('least_squares') == ('least_squares', None)
    assert r('differential_evolution') == ('differential_evolution', None)
    assert r('powell') == ('scalar', 'Powell') and r('nelder') == ('scalar', 'Nelder-Mead')
    assert r('Nelder-Mead') == ('scalar', 'Nelder-Mead') and r('L-BFGS-B') == ('scalar', 'L-BFGS-B')
    assert r('lbfgsb') == ('scalar', 'L-BFGS-B') and r('cobyla') == ('scalar', 'COBYLA')
    # the docstring's "levenberg-marquardt" (library.py:699) matches nothing in lmfit's tables: scalar_minimize's default
    assert r('levenberg-marquardt') == ('scalar', 'Nelder-Mead')
    for name in ('emcee', 'basinhopping', 'newton', 'dogleg', 'trust-krylov'):
        with pytest.raises(NotImplementedError):
            r(name)


def test_bounds_transform_round_trip_and_range():
    pair = fitting.BoundedPair([300.0, 40.0], [60.0, 8.0])
    assert pair.lo.tolist() == [240.0, 32.0] and pair.hi.tolist() == [360.0, 48.0]
    v = np.array([[241.0, 47.5], [300.0, 40.0], [359.9, 32.1]])
    np.testing.assert_allclose(pair.from_internal(pair.to_internal(v)), v, rtol=1e-13)
    x = np.linspace(-50.0, 50.0, 1001)                    # any internal value lands inside the bounds
    inside = pair.from_internal(np.stack([x, x[::-1]], axis=1))
    assert np.all(inside >= pair.lo) and np.all(inside <= pair.hi)
    assert pair.to_internal(pair.value).tolist() == [0.0, 0.0]
    with pytest.raises(ValueError):
        fitting.BoundedPair([300.0, 0.0], [60.0, 0.0])


@pytest.mark.parametrize("family,name,tol", [("leastsq", None, 1e-5), ("least_squares", None, 1e-5),
                                             ("scalar", "Nelder-Mead", 1e-2), ("scalar", "Powell", 1e-3),
                                             ("scalar", "L-BFGS-B", 1e-2), ("scalar", "COBYLA", 0.5),
                                             ("differential_evolution", None, 0.2)])
def test_every_optimiser_family_finds_the_minimum_of_a_synthetic_trace(family, name, tol):
    f = np.linspace(2.0, 9.0, 29)
    truth = np.array([287.0, 43.5])

    def model(nodes):
        hm, bb = nodes[:, :1], nodes[:, 1:]
        return hm * (1.0 + 0.05 * np.sin(f)) + bb * np.sqrt(f) + 1e-3 * (hm - 250.0) * bb
    obs = model(truth[None, :])[0]
    batches = []

    def residuals(nodes):
        nodes = np.atleast_2d(np.asarray(nodes, dtype=np.float64))
        assert nodes.shape[1] == 2
        batches.append(nodes.shape[0])
        return obs - model(nodes)
    pair = fitting.BoundedPair([300.0, 40.0], [60.0, 8.0])
    got = fitting._local_search(residuals, pair, family, name)
    assert got.shape == (2,) and np.all(np.abs(got - truth) < tol), (family, name, got)
    if family == "leastsq":
        assert set(batches) == {1, 3}                     # single evaluations and Jacobian batches of n + 1 rows
    if family == "differential_evolution":
        assert max(batches) >= 30


def test_a_nan_residual_raises_as_lmfit_does_by_default():
    pair = fitting.BoundedPair([300.0, 40.0], [60.0, 8.0])

    def residuals(nodes):
        return np.full((np.atleast_2d(nodes).shape[0], 5), np.nan)
    for family, name in (("leastsq", None), ("least_squares", None), ("scalar", "Powell")):
        with pytest.raises(ValueError, match="NaN values detected"):
            fitting._local_search(residuals, pair, family, name)


def test_the_default_builder_is_pyiri_and_says_so_when_it_is_absent():
    """model_VH / residual_VH / minimize_parameters keep the reference's positional signatures; what the reference
    builds with PyIRI (library.py:557-586) comes from pyiri_edp_builder by default - PyIRI is absent here, so the call
    must fail with an ImportError that names the keyword to pass instead (before anything touches the GPU)."""
    import inspect
    ref_args = ["F2", "F1", "E", "f_in0", "vh_obs0", "alt", "b_mag", "b_psi", "method", "percent_sigma", "step", "mode",
                "n_points", "bottom_type"]
    sig = inspect.signature(fitting.minimize_parameters)
    assert [p for p, v in sig.parameters.items() if v.kind is v.POSITIONAL_OR_KEYWORD] == ref_args      # library.py:672-674
    assert [p for p, v in inspect.signature(fitting.model_VH).parameters.items() if v.kind is v.POSITIONAL_OR_KEYWORD] == \
        ["F2", "F1", "E", "f_in", "alt", "b_mag", "b_psi", "mode", "n_points", "bottom_type"]           # library.py:512-513
    assert [p for p, v in inspect.signature(fitting.residual_VH).parameters.items() if v.kind is v.POSITIONAL_OR_KEYWORD] == \
        ["params", "F2_init", "F1_init", "E_init", "f_in", "vh_obs", "alt", "b_mag", "b_psi", "mode", "n_points",
         "bottom_type"]                                                                                  # library.py:595-596
    try:
        import PyIRI  # noqa: F401
        pytest.skip("PyIRI is installed here")
    except ImportError:
        pass
    one = lambda v: np.array([[[v]]])                                           # noqa: E731
    F2 = {"Nm": one(1e12), "hm": one(300.0), "B_bot": one(40.0)}
    F1, E = {"P": one(0.5)}, {"hm": one(110.0)}
    alt = np.arange(80.0, 500.0)
    f = np.arange(2.0, 8.0, 0.5)
    with pytest.raises(ImportError, match="edp_builder"):
        fitting.model_VH(F2, F1, E, f, alt, np.full(alt.size, 4e-5), np.full(alt.size, 30.0))
    with pytest.raises(ImportError, match="edp_builder"):
        fitting.residual_VH({"NmF2": 1e12, "hmF2": 300.0, "B_bot": 40.0}, F2, F1, E, f, np.full(f.size, 250.0), alt,
                            np.full(alt.size, 4e-5), np.full(alt.size, 30.0))
    with pytest.raises(ImportError, match="edp_builder"):
        fitting.minimize_parameters(F2, F1, E, f, np.full(f.size, 250.0), alt, np.full(alt.size, 4e-5),
                                    np.full(alt.size, 30.0))
