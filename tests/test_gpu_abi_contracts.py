"""Contracts of the C ABI that only show on a GPU: segment validation, NaN rows of uncovered output,
the caller's current device, stream hand-back, and the device-side profile-index check of the tracers."""

import ctypes

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _worklist(ctx, g, segs, n_rows_out, n_prof=24):
    from pyrayhf_amd import _native, library
    f = np.ascontiguousarray(g["freq"][:32])
    d, b, p = (np.ascontiguousarray(g[k][:n_prof]) for k in ("den", "bmag", "bpsi"))
    grids = [library._multiplier(s.n_points) for s in segs]
    mult = np.ascontiguousarray(np.concatenate(grids))
    out = np.full((n_rows_out, f.size), -7.0)
    rc = ctx.vfo_worklist(f.ctypes.data, f.size, d.ctypes.data, b.ctypes.data, p.ctypes.data, g["alt"].ctypes.data,
                          n_prof, d.shape[1], d.shape[1], 0, mult.ctypes.data, mult.size, segs, out.ctypes.data, 0)
    return rc, out, _native.last_error()


def test_worklist_rejects_misaligned_and_overlapping_output_ranges():
    from pyrayhf_amd import _native
    g = load_golden("g5_chapman64.npz")
    ctx = _native.host_context(0)
    S = _native.Segment
    # an offset that is not a multiple of n_freq would under-size the staged output
    rc, _, msg = _worklist(ctx, g, [S(0, 4, _native.MODE_X, 200, 0, 5)], 8)
    assert rc == _native.EINVAL and "multiple of n_freq" in msg
    # two segments writing rows 2..5 and 4..7
    rc, _, msg = _worklist(ctx, g, [S(0, 4, _native.MODE_X, 200, 0, 2 * 32), S(4, 8, _native.MODE_X, 200, 200, 4 * 32)], 8)
    assert rc == _native.EINVAL and "same output rows" in msg
    # disjoint ranges with a gap: rows 0..3 and 6..9 written, rows 4..5 come back as NaN (not arena bytes)
    rc, out, msg = _worklist(ctx, g, [S(0, 4, _native.MODE_X, 200, 0, 0), S(4, 8, _native.MODE_X, 200, 200, 6 * 32)], 10)
    assert rc == _native.OK, msg
    assert np.isnan(out[4:6]).all() and np.isfinite(out[:4]).any() and np.isfinite(out[6:]).any()
    assert not (out == -7.0).any()


def test_calls_leave_the_current_device_alone():
    """every entry point used to hipSetDevice(ctx->device) and leave it set.  On a one-GPU box the
    observable part is that torch's current device and stream still work after calls on another thread's
    context; with two GPUs the current device itself is checked."""
    import torch
    from pyrayhf_amd import _native, library
    g = load_golden("g1_basic.npz")
    n = _native.device_count()
    before = torch.cuda.current_device()
    for dev in range(min(n, 2)):
        library.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 50, device=dev)
        assert torch.cuda.current_device() == before
    if n >= 2:
        torch.cuda.set_device(1)
        library.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 50, device=0)
        t = torch.ones(4, device="cuda")                       # lands on the device the caller selected
        assert torch.cuda.current_device() == 1 and t.device.index == 1
        torch.cuda.set_device(before)
    # raw HIP view of the same fact
    hip = ctypes.CDLL("libamdhip64.so")
    cur = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == torch.cuda.current_device()


def test_host_call_after_a_torch_call_on_a_dead_side_stream():
    """The torch path borrows torch's current stream; a later NumPy-path call must run on the context's own
    stream again, even when the borrowed stream no longer exists."""
    import gc
    import torch
    from pyrayhf_amd import library
    g = load_golden("g5_chapman64.npz")
    dev = torch.device("cuda:0")
    t = [torch.as_tensor(g[k], device=dev) for k in ("freq", "den", "bmag", "bpsi", "alt")]
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        a = library.vertical_forward_operator(*t, "X", 2000)
    side.synchronize()
    want = a.cpu().numpy()
    del side
    gc.collect()
    torch.cuda.empty_cache()
    host = library.vertical_forward_operator(*(g[k] for k in ("freq", "den", "bmag", "bpsi", "alt")), "X", 2000)
    assert np.array_equal(host, want, equal_nan=True)


def test_tracer_reports_a_bad_device_resident_profile_index():
    import torch
    from pyrayhf_amd import _native
    g = load_golden("g8_snell.npz")
    prof = {k: torch.as_tensor(np.ascontiguousarray(g[f"gauss_{k}"]).reshape(1, -1), device="cuda") for k in ("den", "bmag", "bpsi")}
    alt = torch.as_tensor(g["gauss_alt"], device="cuda")
    n_alt = alt.numel()
    f = torch.full((3,), 5e6, dtype=torch.float64, device="cuda")
    e = torch.full((3,), 45.0, dtype=torch.float64, device="cuda")
    out = torch.zeros((3, 8), dtype=torch.float64, device="cuda")
    ctx = _native.host_context(0)
    for idx, want_rc in (([0, 0, 0], _native.OK), ([0, 7, 0], _native.EINVAL), ([0, 0, -1], _native.EINVAL)):
        pi = torch.tensor(idx, dtype=torch.int64, device="cuda")
        rc = ctx.snell_cartesian(f.data_ptr(), e.data_ptr(), pi.data_ptr(), 3, prof["den"].data_ptr(), prof["bmag"].data_ptr(),
                                 prof["bpsi"].data_ptr(), alt.data_ptr(), 1, n_alt, 0, _native.MODE_O, out.data_ptr(), 0, 0, 0,
                                 _native.FLAG_DEVICE_PTRS)
        assert rc == want_rc, (idx, rc, _native.last_error())
        o = out.cpu().numpy()
        assert np.isfinite(o[0, 0])                                   # the valid ray is traced either way
        if want_rc != _native.OK:
            bad = [i for i, v in enumerate(idx) if v != 0]
            assert np.isnan(o[bad, 0]).all() and "profile_index" in _native.last_error()


def test_fan_reports_bad_device_resident_groups_and_profile_indices():
    """prhf_snell_fan_f64 on device-resident arrays cannot check ray_group / the groups' profile indices on the host: the
    kernels do (snell_tables_kernel leaves the group's error bits, snell_ray_table posts the status), the call reports
    it at its synchronisation, the rays concerned come back NaN and the others are traced - on both geometries."""
    import torch
    from pyrayhf_amd import _native
    g = load_golden("g8_snell.npz")
    prof = {k: torch.as_tensor(np.ascontiguousarray(g[f"gauss_{k}"]).reshape(1, -1), device="cuda") for k in ("den", "bmag", "bpsi")}
    alt = torch.as_tensor(g["gauss_alt"], device="cuda")
    n_alt = alt.numel()
    gf = torch.tensor([5e6, 6e6], dtype=torch.float64, device="cuda")
    e = torch.full((4,), 45.0, dtype=torch.float64, device="cuda")
    ctx = _native.host_context(0)
    cases = (([0, 1, 1, 0], [0, 0], _native.OK, [], ""),
             ([0, 2, 1, 0], [0, 0], _native.EINVAL, [1], "ray_group"),
             ([0, -1, 1, 0], [0, 0], _native.EINVAL, [1], "ray_group"),
             ([0, 1, 1, 0], [0, 5], _native.EINVAL, [1, 2], "profile_index"))
    for geometry in (0, 1):
        for groups, gprof, want_rc, bad, word in cases:
            out = torch.zeros((4, 8), dtype=torch.float64, device="cuda")
            rg = torch.tensor(groups, dtype=torch.int64, device="cuda")
            gp = torch.tensor(gprof, dtype=torch.int64, device="cuda")
            rc = ctx.snell_fan(geometry, gf.data_ptr(), gp.data_ptr(), 2, rg.data_ptr(), e.data_ptr(), 4, prof["den"].data_ptr(),
                               prof["bmag"].data_ptr(), prof["bpsi"].data_ptr(), alt.data_ptr(), 1, n_alt, 0, _native.MODE_O,
                               6371.0, 1.0, 200.0, 400, out.data_ptr(), 0, 0, 0, _native.FLAG_DEVICE_PTRS)
            assert rc == want_rc, (geometry, groups, gprof, rc, _native.last_error())
            o = out.cpu().numpy()
            good = [i for i in range(4) if i not in bad]
            assert np.isfinite(o[good, 0]).all(), (geometry, groups, gprof, o[:, 0])
            if bad:
                assert np.isnan(o[bad, 0]).all() and word in _native.last_error(), (geometry, groups, gprof, _native.last_error())


def test_recent_kernel_ms_reports_every_launch_of_an_unsynchronised_series():
    """prhf_recent_kernel_ms: five launches enqueued back to back, their device times read afterwards with one
    synchronisation; the newest equals prhf_last_kernel_ms; the context remembers 64."""
    import torch
    from pyrayhf_amd import library, _native
    g = load_golden("g5_chapman64.npz")
    dev = torch.device("cuda:0")
    t = {k: torch.as_tensor(g[k], device=dev) for k in ("freq", "den", "bmag", "bpsi", "alt")}
    out = torch.empty((64, g["freq"].size), dtype=torch.float64, device=dev)
    sizes = (200, 2000, 20000, 2000, 200)
    for n in sizes:
        library.vertical_forward_operator(t["freq"], t["den"], t["bmag"], t["bpsi"], t["alt"], "X", n, sync=False, out=out)
    ctx = _native.context(0)
    ms = ctx.recent_kernel_ms(len(sizes))
    assert len(ms) == len(sizes) and all(m > 0.0 for m in ms)
    assert ms[2] > ms[0] and ms[2] > ms[4]                      # the 20000-point launch is the long one
    assert ms[-1] == ctx.last_kernel_ms()
    assert len(ctx.recent_kernel_ms(2)) == 2 and ctx.recent_kernel_ms(2) == ms[-2:]
    for _ in range(70):
        library.vertical_forward_operator(t["freq"], t["den"], t["bmag"], t["bpsi"], t["alt"], "X", 200, sync=False, out=out)
    assert len(ctx.recent_kernel_ms(1000)) == 64
    assert library.recent_kernel_ms(3) == ctx.recent_kernel_ms(3)


def test_host_buffer_calls_are_timed_on_request_only():
    """Option `timing`: a synchronous host-buffer call records its two timing events only when asked to;
    launches on device pointers always do."""
    import torch
    from pyrayhf_amd import _native, library, synth
    alt, den, bmag, bpsi = synth.chapman_profiles(2, 3)
    freq = synth.sounder_frequencies(1)
    ctx = _native.Context(0)                      # a fresh context: nothing timed yet
    try:
        out = np.empty((1, freq.size))
        m = np.ascontiguousarray(library.smooth_nonuniform_grid(0, 1, 200, 10.0))
        args = (freq.ctypes.data, freq.size, den[0].ctypes.data, bmag[0].ctypes.data, bpsi[0].ctypes.data, alt.ctypes.data,
                1, alt.size, alt.size, 0, m.ctypes.data, 200, _native.MODE_X, out.ctypes.data, 0)
        _native.raise_for(ctx.vfo_batch(*args))
        with pytest.raises(Exception, match="no launch has been timed"):
            ctx.last_kernel_ms()
        ctx.set_option("timing", 1)
        untimed = out.copy()
        _native.raise_for(ctx.vfo_batch(*args))
        assert 0.0 < ctx.last_kernel_ms() < 5.0 and np.array_equal(out, untimed, equal_nan=True)
    finally:
        ctx.close()
