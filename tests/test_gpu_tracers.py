"""Stratified Snell's-law tracer on the GPU against the reference's own outputs (fixture G8,
trace_ray_cartesian_snells run by oracle/gen_golden.py) and its structural test."""

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _midpoint_against_the_reference(x_mid, z_mid, x_mid_ref, z_mid_ref, x_ref, z_ref, rtol):
    """The midpoint rule (reference library.py:1248-1252 / :1691-1695; prhf_snell.inc, in front of snell_ray): on the
    mirrored path the running length behind the last up-leg segment IS half the path, so in exact arithmetic the
    reference's search returns that segment's first node, the one before the apex - which is what the kernels return,
    to the path nodes' tolerance.  The reference compares two roundings of the same number and lands on that node or
    on the apex.  Returns whether the two agree; where they do the values are held to `rtol`."""
    apex = x_ref.size // 2
    assert apex >= 1
    assert abs(x_mid - x_ref[apex - 1]) <= rtol * abs(x_ref[apex - 1]) + 1e-9
    assert abs(z_mid - z_ref[apex - 1]) <= max(rtol / 10, 1e-13) * abs(z_ref[apex - 1]) + 1e-12
    on = [abs(x_mid_ref - x_ref[j]) <= 1e-14 * abs(x_ref[j]) and z_mid_ref == z_ref[j] for j in (apex - 1, apex)]
    assert on[0] or on[1], (x_mid_ref, z_mid_ref, x_ref[apex - 1:apex + 1], z_ref[apex - 1:apex + 1])
    if on[0]:
        assert abs(x_mid - x_mid_ref) <= rtol * abs(x_mid_ref) + 1e-9
        assert abs(z_mid - z_mid_ref) <= max(rtol / 10, 1e-13) * abs(z_mid_ref) + 1e-12
    return bool(on[0])


# The per-ray launch's two arithmetic settings (prhf_snell.inc, REDUCED): the reference's operation order at every level
# holds the reference-run rays to 1e-12 (flat) / 1e-11 (spherical); the default - the reduced algebra where a level is
# far from reflection and from the ray's turning point - to 1e-10 (measured: see test_default_arithmetic_against_...).
TIERS = {"faithful": (1e-12, 1e-11), "default": (1e-10, 1e-10)}


def _math(tier):
    from pyrayhf_amd import library
    return library.MATH_FAITHFUL if tier == "faithful" else None


@pytest.mark.parametrize("tier", ["faithful", "default"])
@pytest.mark.parametrize("name", ["gauss", "day"])
def test_batch_against_reference_rays(name, tier):
    from pyrayhf_amd import tracers
    rtol = TIERS[tier][0]
    g = load_golden("g8_snell.npz")
    prof = [g[f"{name}_{k}"] for k in ("alt", "den", "bmag", "bpsi")]
    rays, want, offs = g[f"{name}_rays"], g[f"{name}_scalars"], g[f"{name}_offsets"]
    midpoints = []
    for mode_i, mode in enumerate("OX"):
        sel = np.nonzero(rays[:, 0] == mode_i)[0]
        r = tracers.trace_rays_cartesian_snells(rays[sel, 1], rays[sel, 2], *prof, mode, return_paths=True, math=_math(tier))
        w = want[sel]
        traced = np.isfinite(w[:, 0])
        assert np.array_equal(np.isfinite(r["group_path_km"]), traced)           # same rays turn
        assert np.all(r["n_path"][~traced] == 0)
        np.testing.assert_allclose(r["group_path_km"][traced], w[traced, 0], rtol=rtol)
        np.testing.assert_allclose(r["group_delay_sec"][traced], w[traced, 1], rtol=rtol)
        gr = w[traced, 4]
        np.testing.assert_allclose(r["ground_range_km"][traced][np.isfinite(gr)], gr[np.isfinite(gr)], rtol=rtol,
                                   atol=1e-12)
        assert np.array_equal(np.isnan(r["ground_range_km"][traced]), np.isnan(gr))
        for k, i in enumerate(sel):
            n = offs[i + 1] - offs[i]
            assert r["n_path"][k] == n
            if n == 0:
                continue
            x_ref, z_ref = g[f"{name}_x"][offs[i]:offs[i + 1]], g[f"{name}_z"][offs[i]:offs[i + 1]]
            np.testing.assert_allclose(r["x"][k, :n], x_ref, rtol=rtol, atol=1e-10)
            np.testing.assert_allclose(r["z"][k, :n], z_ref, rtol=max(rtol / 10, 1e-13), atol=1e-12)
            assert np.all(np.isnan(r["x"][k, n:]))
            apex = n // 2
            np.testing.assert_allclose(r["z_turn_km"][k], z_ref[apex], rtol=max(rtol / 10, 1e-13))
            np.testing.assert_allclose(r["x_turn_km"][k], x_ref[apex], rtol=rtol, atol=1e-10)
            midpoints.append(_midpoint_against_the_reference(r["x_midpoint"][k], r["z_midpoint"][k], w[k, 2], w[k, 3],
                                                             x_ref, z_ref, rtol))
    # the reference's own choice between the two nodes (its rounding decides): the node before the apex for 84 of
    # the 127 rays of G8, the apex for 43; where it is the node before, the values agree to the paths' 1e-12
    assert sum(midpoints) >= 0.6 * len(midpoints), (sum(midpoints), len(midpoints))


def test_single_ray_dict_like_the_reference():
    # reference test_core.py:724-768
    from pyrayhf_amd import tracers
    alt_km = np.linspace(0, 600, 200)
    Ne = 1e12 * np.exp(-(alt_km - 250) ** 2 / (2 * 60 ** 2))
    res = tracers.trace_ray_cartesian_snells(f0_Hz=10e6, elevation_deg=45.0, alt_km=alt_km, Ne=Ne,
                                             Babs=np.full_like(alt_km, 4e-5), bpsi=np.full_like(alt_km, 45.0), mode="O")
    assert {"x", "z", "group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint", "ground_range_km"} <= set(res)
    assert np.all(np.isfinite(res["x"])) and np.all(np.isfinite(res["z"]))
    assert res["group_path_km"] > 0 and res["group_delay_sec"] > 0 and res["ground_range_km"] > 0
    z = res["z"]
    assert np.isclose(z[0], 0.0, atol=1e-3) and np.nanmax(z) > 50.0 and np.isclose(z[-1], 0.0, atol=1e-2)
    # a ray that escapes: every entry NaN
    esc = tracers.trace_ray_cartesian_snells(30e6, 80.0, alt_km, Ne, np.full_like(alt_km, 4e-5),
                                             np.full_like(alt_km, 45.0), "O")
    assert all(np.all(np.isnan(v)) for v in esc.values())
    with pytest.raises(ValueError, match="Mode must be O or X"):
        tracers.trace_ray_cartesian_snells(10e6, 45.0, alt_km, Ne, Ne, Ne, "Q")


def test_rays_over_several_profiles_match_single_profile_calls():
    from pyrayhf_amd import tracers
    g = load_golden("g5_chapman64.npz")
    f = np.array([4e6, 6e6, 8e6, 5e6, 7e6, 3e6])
    e = np.array([30.0, 50.0, 70.0, 80.0, 20.0, 60.0])
    idx = np.array([0, 3, 5, 3, 1, 0])
    many = tracers.trace_rays_cartesian_snells(f, e, g["alt"], g["den"][:6], g["bmag"][:6], g["bpsi"][:6], "X",
                                               profile_index=idx)
    for k in range(f.size):
        one = tracers.trace_rays_cartesian_snells(f[k], e[k], g["alt"], g["den"][idx[k]], g["bmag"][idx[k]],
                                                  g["bpsi"][idx[k]], "X")
        for key in ("group_path_km", "group_delay_sec", "ground_range_km", "z_turn_km"):
            assert np.array_equal(many[key][k:k + 1], one[key], equal_nan=True), (k, key)


@pytest.mark.parametrize("tier", ["faithful", "default"])
@pytest.mark.parametrize("name", ["gauss", "day"])
def test_spherical_batch_against_reference_rays(name, tier):
    """trace_ray_spherical_snells (reference library.py:1460-1713) run by oracle/gen_golden.py: fixture G9."""
    from pyrayhf_amd import tracers
    rtol = TIERS[tier][1]
    g = load_golden("g9_snell_spherical.npz")
    p = load_golden("g8_snell.npz")
    prof = [p[f"{name}_{k}"] for k in ("alt", "den", "bmag", "bpsi")]
    rays, want, offs = g[f"{name}_rays"], g[f"{name}_scalars"], g[f"{name}_offsets"]
    midpoints = []
    for mode_i, mode in enumerate("OX"):
        sel = np.nonzero(rays[:, 0] == mode_i)[0]
        r = tracers.trace_rays_spherical_snells(rays[sel, 1], rays[sel, 2], *prof, mode, return_paths=True, math=_math(tier))
        w = want[sel]
        traced = np.isfinite(w[:, 0])
        assert np.array_equal(np.isfinite(r["group_path_km"]), traced)
        np.testing.assert_allclose(r["group_path_km"][traced], w[traced, 0], rtol=rtol)
        np.testing.assert_allclose(r["group_delay_sec"][traced], w[traced, 1], rtol=rtol)
        gr = w[traced, 4]
        fin = np.isfinite(gr)
        np.testing.assert_allclose(r["ground_range_km"][traced][fin], gr[fin], rtol=rtol, atol=1e-10)
        for k, i in enumerate(sel):
            n = offs[i + 1] - offs[i]
            assert r["n_path"][k] == n
            if n:
                x_ref, z_ref = g[f"{name}_x"][offs[i]:offs[i + 1]], g[f"{name}_z"][offs[i]:offs[i + 1]]
                np.testing.assert_allclose(r["x"][k, :n], x_ref, rtol=rtol, atol=1e-9)
                np.testing.assert_allclose(r["z"][k, :n], z_ref, rtol=max(rtol / 10, 1e-13), atol=1e-12)
                midpoints.append(_midpoint_against_the_reference(r["x_midpoint"][k], r["z_midpoint"][k], w[k, 2],
                                                                 w[k, 3], x_ref, z_ref, rtol))
    assert sum(midpoints) >= 0.5 * len(midpoints), (sum(midpoints), len(midpoints))


def test_spherical_single_ray_and_flat_limit():
    """Structure of the reference's dict, and its flat-Earth limit test (reference test_core.py:843-887):
    with a huge Earth radius the spherical tracer approaches the flat-Earth one."""
    from pyrayhf_amd import tracers
    alt_km = np.linspace(0, 600, 200)
    Ne = 1e12 * np.exp(-(alt_km - 250) ** 2 / (2 * 60 ** 2))
    B, psi = np.full_like(alt_km, 4e-5), np.full_like(alt_km, 45.0)
    sph = tracers.trace_ray_spherical_snells(10e6, 45.0, alt_km, Ne, B, psi, "O")
    assert np.all(np.isfinite(sph["x"])) and sph["group_path_km"] > 0 and sph["ground_range_km"] > 0
    assert np.isclose(sph["z"][0], 0.0, atol=1e-3) and np.isclose(sph["z"][-1], 0.0, atol=1e-3)
    flat = tracers.trace_ray_cartesian_snells(10e6, 50.0, alt_km, Ne, B, psi, "O")
    big = tracers.trace_ray_spherical_snells(10e6, 50.0, alt_km, Ne, B, psi, "O", R_E=6371e9)
    for key in ("group_path_km", "group_delay_sec", "ground_range_km"):      # reference test_core.py:878-881
        v_cart, v_sph = flat[key], big[key]
        assert abs(v_cart - v_sph) / max(abs(v_cart), abs(v_sph)) < 0.03, key
    assert np.nanmax(flat["z"]) > 100.0 and np.nanmax(big["z"]) > 100.0
    for kw in ({"R_E": np.inf}, {"R_E": -6371.0}, {"R_E": np.nan}, {"dz_target_km": 0.0}, {"max_substeps": 0}):
        with pytest.raises(ValueError, match="spherical tracer controls"):
            tracers.trace_ray_spherical_snells(10e6, 50.0, alt_km, Ne, B, psi, "O", **kw)
        with pytest.raises(ValueError, match="spherical tracer controls"):
            tracers.trace_fan_spherical_snells(np.array([10e6]), np.array([50.0]), alt_km, Ne, B, psi, "O", **kw)
    esc = tracers.trace_ray_spherical_snells(30e6, 80.0, alt_km, Ne, B, psi, "O")
    assert set(esc) == {"x", "z", "group_path_km", "group_delay_sec", "x_midpoint", "z_midpoint", "ground_range_km"}
    assert all(np.isnan(v) for v in esc.values())


@pytest.mark.parametrize("spherical", [False, True])
def test_fan_shares_the_levels_and_changes_nothing(spherical):
    """prhf_snell_fan_f64: the refractive-index levels once per (profile, frequency), read by every elevation of the
    fan.  Same levels, same bracket, same up-leg increments as the per-ray call - whose kernel (round 4) makes ONE pass
    over the levels and takes the mirrored half of the path by symmetry, where the fan kernel, with the table in reach,
    walks the up-leg a second time: the two agree to 1e-13 in every scalar and path node (measured 1e-15), the same
    rays turn, the paths have the same nodes, the midpoint is the same node (the one before the apex).
    Through the per-ray call's test both are held to the reference's rays (fixtures G8 / G9 are fans)."""
    from pyrayhf_amd import tracers
    g = load_golden("g8_snell.npz")
    for name in ("gauss", "day"):
        prof = [g[f"{name}_{k}"] for k in ("alt", "den", "bmag", "bpsi")]
        freqs = np.array([2.0, 3.5, 5.0, 7.0, 9.0, 10.0, 12.5, 16.0]) * 1e6
        elevs = np.array([5.0, 20.0, 45.0, 70.0, 85.0, 89.9, 90.0])
        for mode in "OX":
            fan_fn = tracers.trace_fan_spherical_snells if spherical else tracers.trace_fan_cartesian_snells
            ray_fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
            fan = fan_fn(freqs, elevs, *prof, mode, return_paths=True)
            ff, ee = np.meshgrid(freqs, elevs, indexing="ij")
            # (the fan's level tables are in the reference's operation order: so is the per-ray call it is held against)
            rays = ray_fn(ff.ravel(), ee.ravel(), *prof, mode, return_paths=True, math=_math("faithful"))
            for key in ("group_path_km", "group_delay_sec", "ground_range_km", "x_turn_km", "z_turn_km", "x_midpoint",
                        "z_midpoint", "n_path", "x", "z"):
                assert fan[key].shape[:2] == (freqs.size, elevs.size)
                got, want = fan[key].reshape(rays[key].shape), rays[key]
                assert np.array_equal(np.isnan(got), np.isnan(want)), (name, mode, key)
                if key == "n_path":
                    assert np.array_equal(got, want), (name, mode, key)
                else:
                    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-12, equal_nan=True, err_msg=f"{name} {mode} {key}")
            # midpoints: the node before the apex in both kernels (a path node: bit for bit the path array's entry)
            n_path = rays["n_path"]
            for k in np.nonzero(n_path > 0)[0]:
                apex = int(n_path[k]) // 2
                assert rays["x_midpoint"][k] == rays["x"][k, apex - 1] and rays["z_midpoint"][k] == rays["z"][k, apex - 1]
            assert np.isfinite(fan["group_path_km"]).any() and np.isnan(fan["group_path_km"]).any()
    # several profiles: (P, F, E)
    two = [np.stack([g["gauss_den"], 0.5 * g["gauss_den"]]), np.stack([g["gauss_bmag"]] * 2), np.stack([g["gauss_bpsi"]] * 2)]
    fan = tracers.trace_fan_cartesian_snells(np.array([5e6, 7e6]), np.array([20.0, 60.0, 80.0]), g["gauss_alt"], *two, "O")
    assert fan["group_path_km"].shape == (2, 2, 3)
    one = tracers.trace_fan_cartesian_snells(np.array([5e6, 7e6]), np.array([20.0, 60.0, 80.0]), g["gauss_alt"],
                                             two[0][1], two[1][1], two[2][1], "O")
    assert np.array_equal(fan["group_path_km"][1], one["group_path_km"], equal_nan=True)


@pytest.mark.parametrize("spherical", [False, True])
def test_negative_density_raises_as_in_the_reference(spherical):
    """den2freq raises on a negative density (reference library.py:93-94, the message pinned by test_core.py), and
    the tracers call it on the whole column (:1184 / :1566): the per-ray and the grouped launch report it - from the
    kernels, which look at the columns; a call on columns without one is not disturbed by the call before."""
    from pyrayhf_amd import synth, tracers
    alt, den, bmag, bpsi = synth.chapman_profiles(3, 5)
    bad = den.copy(); bad[1, 400] = -1.0                         # (above every turning point: it is the column that counts)
    f = np.array([4e6, 6e6, 8e6]); e = np.array([30.0, 50.0, 70.0])
    ray_fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
    fan_fn = tracers.trace_fan_spherical_snells if spherical else tracers.trace_fan_cartesian_snells
    one_fn = tracers.trace_ray_spherical_snells if spherical else tracers.trace_ray_cartesian_snells
    with pytest.raises(ValueError, match="Density must be non-negative"):
        ray_fn(f, e, alt, bad, bmag, bpsi, "O", profile_index=np.array([0, 1, 2]))
    with pytest.raises(ValueError, match="Density must be non-negative"):
        fan_fn(f, e, alt, bad, bmag, bpsi, "O")
    with pytest.raises(ValueError, match="Density must be non-negative"):
        one_fn(6e6, 45.0, alt, bad[1], bmag[1], bpsi[1], "X")
    good = ray_fn(f, e, alt, den, bmag, bpsi, "O", profile_index=np.array([0, 1, 2]))
    assert np.isfinite(good["group_path_km"]).any()
    # rays that do not touch the bad column are not an error
    ok = ray_fn(f, e, alt, bad, bmag, bpsi, "O", profile_index=np.array([0, 2, 2]))
    assert np.array_equal(ok["group_path_km"][0], good["group_path_km"][0])
    assert fan_fn(f, e, alt, den, bmag, bpsi, "O")["group_path_km"].shape == (3, 3, 3)


@pytest.mark.parametrize("spherical", [False, True])
def test_ray_queues_give_every_ray_its_result_whatever_the_count(spherical):
    """The per-ray launch is persistent: wavefronts draw their rays four at a time from eight queues, one per slice of
    the rays (prhf_snell.inc snell_dispatch).  Ray counts that are multiples of nothing, below and above the number of
    slices, the batch and the resident wavefronts: every ray comes out as in the launch of all 30 011 (the reference's
    operation order, where the per-profile table - taken from 4 rays per profile on - changes no bit)."""
    from pyrayhf_amd import library, synth, tracers
    alt, den, bmag, bpsi = synth.chapman_profiles(16, 3)
    rng = np.random.default_rng(9)
    R = 30011
    f = rng.uniform(2e6, 14e6, R); e = rng.uniform(1.0, 90.0, R); idx = rng.integers(0, 16, R)
    fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
    keys = ("group_path_km", "group_delay_sec", "ground_range_km", "x_midpoint", "z_midpoint", "n_path")
    whole = fn(f, e, alt, den, bmag, bpsi, "O", profile_index=idx, math=library.MATH_FAITHFUL)
    assert 0.3 < np.isfinite(whole["group_path_km"]).mean() < 0.95
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 65, 257, 4099, 20481):
        part = fn(f[:n], e[:n], alt, den, bmag, bpsi, "O", profile_index=idx[:n], math=library.MATH_FAITHFUL)
        for key in keys:
            assert np.array_equal(part[key], whole[key][:n], equal_nan=True), (n, key)
    tail = fn(f[-1777:], e[-1777:], alt, den, bmag, bpsi, "O", profile_index=idx[-1777:], math=library.MATH_FAITHFUL)
    for key in keys:
        assert np.array_equal(tail[key], whole[key][-1777:], equal_nan=True), key


@pytest.mark.parametrize("spherical", [False, True])
def test_fan_bracket_by_running_minimum_on_odd_columns(spherical):
    """The grouped launch finds the reference's bracket (library.py:1085-1093 / :1598-1603: the FIRST pair of consecutive
    finite levels with crit[i] >= p >= crit[i + 1]) as the first entry whose running minimum of the criterion is <= p
    (prhf_snell.inc snell_ray_table), checks the pair against the reference's condition and walks the list when the
    check fails.  Held against the per-ray launch in the reference's operation order, which walks the levels itself:
    a valley that blanks levels between two layers, a layer of vacuum, an unmagnetised column, a column whose
    criterion is NaN at one level (a NaN altitude: the walk's case), elevations at and beyond the ends of [0, 90]."""
    from pyrayhf_amd import library, synth, tracers
    alt, den, bmag, bpsi = synth.chapman_profiles(6, 31)
    z = alt if alt.ndim == 1 else alt[0]
    den = den.copy()
    den[1] += 2.5 * den[1].max() * np.exp(-0.5 * ((z - 110.0) / 6.0) ** 2)      # an E layer denser than the F peak: a blanked band
    den[2, 90:130] = 0.0                                                          # vacuum in the column
    bmag = bmag.copy(); bmag[3] = 0.0
    grids = [alt]
    if spherical:
        odd = np.array(alt, dtype=float, copy=True)
        odd[..., 140] = np.nan
        grids.append(odd)
    freqs = np.linspace(1.5e6, 14e6, 26)
    elevs = np.concatenate([np.linspace(0.0, 90.0, 31), [89.99, 90.5, -3.0]])
    fan_fn = tracers.trace_fan_spherical_snells if spherical else tracers.trace_fan_cartesian_snells
    ray_fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
    pp, ff, ee = np.meshgrid(np.arange(6), freqs, elevs, indexing="ij")
    turned = 0
    for grid in grids:
        for mode in "OX":
            fan = fan_fn(freqs, elevs, grid, den, bmag, bpsi, mode, return_paths=True)
            rays = ray_fn(ff.ravel(), ee.ravel(), grid, den, bmag, bpsi, mode, profile_index=pp.ravel(), return_paths=True,
                          math=library.MATH_FAITHFUL)
            for key in ("group_path_km", "group_delay_sec", "ground_range_km", "x_turn_km", "z_turn_km", "x_midpoint",
                        "z_midpoint", "n_path", "x", "z"):
                got, want = fan[key].reshape(rays[key].shape), rays[key]
                assert np.array_equal(np.isnan(got), np.isnan(want)), (mode, key)
                if key == "n_path":
                    assert np.array_equal(got, want), (mode, key)
                else:
                    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-12, equal_nan=True, err_msg=f"{mode} {key}")
            turned += int(np.isfinite(rays["group_path_km"]).sum())
            # every profile has rays that turn and rays that escape
            per_prof = np.isfinite(fan["group_path_km"]).reshape(6, -1)
            assert per_prof.any(axis=1).all() and (~per_prof).any(axis=1).all()
    assert turned > 3000


@pytest.mark.parametrize("spherical", [False, True])
def test_per_profile_level_table_changes_no_bit(spherical):
    """Option snell_table: f_N^2, g_p |B|, sin(psi), cos(psi) of every level once per profile (snell_profile_kernel)
    instead of per ray and level - hoisted, not changed: the rays of the per-ray call, of the grouped call and their
    paths come out bit for bit the same with the table (default when the rays outnumber the profiles four to one),
    without it (0) and with it forced on a launch that would not take it (1 ray per profile) - in the reference's operation
    order; the default arithmetic reads f_N^2 and sin^2(psi) from the table where it has one and forms them itself where
    it has not, which is the same number to the last bits only."""
    from pyrayhf_amd import library, synth, tracers
    alt, den, bmag, bpsi = synth.chapman_profiles(24, 77)
    bmag[3] = 0.0                                            # an unmagnetised column (library.py:201-207)
    rng = np.random.default_rng(5)
    n = 600
    f = rng.uniform(2e6, 14e6, n)
    e = rng.uniform(3.0, 90.0, n)
    idx = rng.integers(0, 24, n)
    ray_fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
    fan_fn = tracers.trace_fan_spherical_snells if spherical else tracers.trace_fan_cartesian_snells
    keys = ("group_path_km", "group_delay_sec", "ground_range_km", "x_turn_km", "z_turn_km", "x_midpoint", "z_midpoint",
            "n_path", "x", "z")
    faithful = library.MATH_FAITHFUL                           # (in the default arithmetic the table changes the last bits:
    try:                                                       #  X = f_N^2 (1 / f^2) from the table's f_N^2 - held to 1e-10 below)
        for mode in "OX":
            got = {}
            for setting in (4.0, 0.0):
                library.set_option("snell_table", setting)
                got[setting] = (ray_fn(f, e, alt, den, bmag, bpsi, mode, profile_index=idx, return_paths=True, math=faithful),
                                fan_fn(np.array([3e6, 6e6, 9e6, 12e6]), np.array([10.0, 45.0, 80.0]), alt, den[:5], bmag[:5],
                                       bpsi[:5], mode, return_paths=True))
            for a, b in zip(got[4.0], got[0.0]):
                for key in keys:
                    assert np.array_equal(a[key], b[key], equal_nan=True), (mode, key)
            assert np.isfinite(got[4.0][0]["group_path_km"]).sum() > 100
            # one ray per profile: not worth a table by default (24 rays < 4 x 24 profiles); forced, the same bits
            few = {}
            for setting in (4.0, 1.0):
                library.set_option("snell_table", setting)
                few[setting] = ray_fn(f[:24], e[:24], alt, den, bmag, bpsi, mode, profile_index=np.arange(24), math=faithful)
            for key in keys[:7]:
                assert np.array_equal(few[4.0][key], few[1.0][key], equal_nan=True), (mode, key)
    finally:
        library.set_option("snell_table", 4.0)


@pytest.mark.parametrize("tier", ["faithful", "default"])
@pytest.mark.parametrize("spherical", [False, True])
def test_random_rays_against_the_oracle(spherical, tier):
    """240 random rays (frequency, elevation, profile; both modes) over seeded Chapman profiles against the NumPy
    restatement of the reference's tracers (oracle/snell_numpy.py, itself held to the reference-run rays of G8 / G9):
    the same rays turn, and path length, group delay, ground range and every path node agree to the fixtures' 1e-12
    (flat) / 1e-11 (spherical).  Covers what the fixtures' two profiles do not: the per-profile level table across 16
    profiles, E-F valleys, oblique and near-vertical rays in one launch."""
    from oracle import snell_numpy as sn
    from pyrayhf_amd import synth, tracers
    alt, den, bmag, bpsi = synth.chapman_profiles(16, 4242)
    rng = np.random.default_rng(99 + int(spherical))
    n = 120
    f = rng.uniform(2e6, 15e6, n)
    e = np.concatenate([rng.uniform(3.0, 88.0, n - 10), rng.uniform(88.0, 90.0, 10)])
    idx = rng.integers(0, 16, n)
    fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
    ofn = sn.trace_spherical if spherical else sn.trace_cartesian
    rtol = TIERS[tier][1 if spherical else 0]
    turned = 0
    midpoints = []
    for mode in "OX":
        got = fn(f, e, alt, den, bmag, bpsi, mode, profile_index=idx, return_paths=True, math=_math(tier))
        for k in range(n):
            with np.errstate(all="ignore"):
                want = ofn(f[k], e[k], alt, den[idx[k]], bmag[idx[k]], bpsi[idx[k]], mode)
            traced = np.isfinite(want["group_path_km"])
            assert np.isfinite(got["group_path_km"][k]) == traced, (mode, k)
            if not traced:
                assert got["n_path"][k] == 0
                continue
            turned += 1
            for key in ("group_path_km", "group_delay_sec"):
                assert abs(got[key][k] - want[key]) <= rtol * abs(want[key]), (mode, k, key, got[key][k], want[key])
            gr = want["ground_range_km"]
            assert np.isnan(got["ground_range_km"][k]) == np.isnan(gr)
            if np.isfinite(gr):
                assert abs(got["ground_range_km"][k] - gr) <= rtol * abs(gr) + 1e-10
            m = want["x"].size
            assert got["n_path"][k] == m
            np.testing.assert_allclose(got["x"][k, :m], want["x"], rtol=rtol, atol=1e-9)
            np.testing.assert_allclose(got["z"][k, :m], want["z"], rtol=rtol, atol=1e-11)
            midpoints.append(_midpoint_against_the_reference(got["x_midpoint"][k], got["z_midpoint"][k],
                                                             want["x_midpoint"], want["z_midpoint"], want["x"],
                                                             want["z"], rtol))
    assert turned > 100
    assert sum(midpoints) >= 0.5 * len(midpoints), (sum(midpoints), len(midpoints))


@pytest.mark.parametrize("spherical", [False, True])
def test_default_arithmetic_against_the_reference_order_on_many_rays(spherical):
    """The per-ray launch's default (reduced algebra where a level is far from reflection and from the ray's turning
    point) against the reference's operation order at every level, on 20 000 random rays over 64 profiles, both modes,
    with and without the per-profile table: the same rays turn, the paths have the same nodes, and path length, group
    delay, ground range and turning point agree to 1e-10 (measured and printed: the worst of each)."""
    from pyrayhf_amd import library, synth, tracers
    alt, den, bmag, bpsi = synth.chapman_profiles(64, 31337)
    rng = np.random.default_rng(7 + int(spherical))
    n = 20000
    f = rng.uniform(2e6, 15e6, n)
    e = np.concatenate([rng.uniform(3.0, 88.0, n - 2000), rng.uniform(88.0, 90.0, 2000)])
    idx = rng.integers(0, 64, n)
    fn = tracers.trace_rays_spherical_snells if spherical else tracers.trace_rays_cartesian_snells
    keys = ("group_path_km", "group_delay_sec", "ground_range_km", "x_turn_km", "z_turn_km", "x_midpoint", "z_midpoint")
    try:
        for table in (4.0, 0.0):
            library.set_option("snell_table", table)
            for mode in "OX":
                ref = fn(f, e, alt, den, bmag, bpsi, mode, profile_index=idx, math=library.MATH_FAITHFUL)
                got = fn(f, e, alt, den, bmag, bpsi, mode, profile_index=idx)
                assert np.array_equal(got["n_path"], ref["n_path"]), (mode, table)
                worst = {}
                for key in keys:
                    a, b = got[key], ref[key]
                    assert np.array_equal(np.isnan(a), np.isnan(b)), (mode, table, key)
                    ok = np.isfinite(b)
                    worst[key] = float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1e-3), initial=0.0))
                    assert worst[key] <= 1e-10, (mode, table, key, worst[key])
                print(f"{'spherical' if spherical else 'flat'} {mode} table={table}: {int(np.isfinite(ref['group_path_km']).sum())} "
                      f"rays turn; worst deviation " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    finally:
        library.set_option("snell_table", 4.0)
