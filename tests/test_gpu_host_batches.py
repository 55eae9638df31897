"""Host-buffer batches: the slab pipeline of prhf_vfo_batch_f64 (uploads and downloads beside the kernels) and the
drop-in call's `devices=` option (one host thread and context per entry, rows cut by shard_bounds).  Both must give
the single-launch values bit for bit, and surface the same errors."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pyrayhf_amd import library
    return library


@pytest.fixture(scope="module")
def batch():
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(1800, 424242)          # 1800 x 620 x 24 B = 26.8 MB of inputs: slabs
    return synth.sounder_frequencies(4)[::2], den, bmag, bpsi, alt


@pytest.mark.parametrize("mode,n", [("X", 2000), ("O", 200), ("X", 200)])
def test_slab_pipeline_changes_nothing(lib, batch, mode, n):
    freq, den, bmag, bpsi, alt = batch
    lib.set_option("host_slabs", 1)
    try:
        whole = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n)
    finally:
        lib.set_option("host_slabs", 3)
    slabs = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n)
    assert np.array_equal(whole, slabs, equal_nan=True)
    assert 0.3 < np.isfinite(slabs).mean() < 0.8
    # per-profile altitudes, strided rows, one shared field row
    alt2 = np.repeat(alt[None, :], den.shape[0], axis=0)
    assert np.array_equal(lib.vertical_forward_operator(freq, den, bmag, bpsi, alt2, mode, n), whole, equal_nan=True)
    wide = np.zeros((den.shape[0], 700))
    wide[:, :620] = den
    assert np.array_equal(lib.vertical_forward_operator(freq, wide[:, :620], bmag, bpsi, alt, mode, n), whole, equal_nan=True)
    one = lib.vertical_forward_operator(freq, den, bmag[7], bpsi[7], alt, mode, n)
    ref = lib.vertical_forward_operator(freq, den, np.repeat(bmag[7:8], den.shape[0], 0), np.repeat(bpsi[7:8], den.shape[0], 0),
                                        alt, mode, n)
    assert np.array_equal(one, ref, equal_nan=True)


def test_slab_pipeline_reports_data_errors(lib, batch):
    freq, den, bmag, bpsi, alt = batch
    bad = den.copy()
    bad[1500, 40] = -1.0                       # in the last slab, below the peak
    with pytest.raises(ValueError, match="Density must be non-negative"):
        lib.vertical_forward_operator(freq, bad, bmag, bpsi, alt, "X", 200)
    with pytest.raises(ValueError, match="mode must be"):
        lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "Z", 200)
    # and the context is usable afterwards
    ok = lib.vertical_forward_operator(freq, den[:100], bmag[:100], bpsi[:100], alt, "X", 200)
    assert np.isfinite(ok).any()


def test_two_contexts_on_one_device_equal_the_single_call(lib, batch):
    """`devices=[0, 0]`: two host threads, two contexts, each its block of rows - the N-GPU path of the drop-in call on
    the one GPU a test box has."""
    freq, den, bmag, bpsi, alt = batch
    for mode, n in (("X", 2000), ("O", 200)):
        single = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n)
        for ids in ([0, 0], [0, 0, 0], "all"):
            got = lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, mode, n, devices=ids)
            assert got.shape == single.shape and np.array_equal(got, single, equal_nan=True), (mode, n, ids)
    # fewer rows than two per device: one launch
    few = lib.vertical_forward_operator(freq, den[:3], bmag[:3], bpsi[:3], alt, "X", 200, devices=[0, 0])
    assert np.array_equal(few, lib.vertical_forward_operator(freq, den[:3], bmag[:3], bpsi[:3], alt, "X", 200), equal_nan=True)
    # an error in one block reaches the caller with the reference's type and message
    bad = den.copy()
    bad[1700, 30] = -5.0
    with pytest.raises(ValueError, match="Density must be non-negative"):
        lib.vertical_forward_operator(freq, bad, bmag, bpsi, alt, "O", 200, devices=[0, 0])
    with pytest.raises(ValueError):
        lib.vertical_forward_operator(freq, den, bmag, bpsi, alt, "O", 200, devices=[])
