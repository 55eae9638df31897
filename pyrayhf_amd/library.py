"""Drop-in for the reference's vertical-ionogram forward operator, on MI355X.

``vertical_forward_operator`` has the signature, positional order, units, output shape and
error behaviour of ``PyRayHF.library.vertical_forward_operator`` (reference
``PyRayHF/library.py:459-509``); the arithmetic runs in libprhf.so's fused HIP kernel
(``pyrayhf_amd/csrc/prhf_kernels.hip``).  Extensions that the reference does not have:

* ``den/bmag/bpsi`` may be 2-D ``(P, N_alt)`` (``alt`` shared 1-D or 2-D): the result is
  ``(P, F)`` and equals a Python loop of single-profile calls;
* inputs may be ``torch`` tensors resident on the GPU (zero-copy, result is a tensor);
* ``vertical_forward_operator_mixed`` evaluates slices with different mode / n_points
  in one launch (BASELINE config 5).

The tiny host-side helpers of the path (constants, unit conversions, the stretched unit
grid) are provided under the reference's names so that callers can switch imports.
"""

from __future__ import annotations

import weakref

import numpy as np

from . import _native, logger

__all__ = [
    "constants", "den2freq", "freq2den", "find_X", "find_Y", "smooth_nonuniform_grid",
    "vertical_to_magnetic_angle", "find_mu_mup", "vertical_forward_operator",
    "vertical_forward_operator_mixed", "last_kernel_ms", "MATH_FAITHFUL", "MATH_FAST", "MATH_AUTO",
]

MATH_FAITHFUL = _native.MATH_FAITHFUL   # reference operation order, IEEE divide / sqrt
MATH_FAST = _native.MATH_FAST           # reduced algebra, rsqrt + Newton, FMA contraction
MATH_AUTO = _native.MATH_AUTO           # default, per slice: 'X' fast; 'O' reference order where 1 - X <= 1e-5, fast elsewhere


# ----------------------------------------------------------------------------------------
# host helpers (reference library.py:40-158, :296-321, :441-456); closed-form, O(N) or less
# ----------------------------------------------------------------------------------------
def constants():
    """(cp [Hz m^1.5], g_p [Hz/T], R_E [km], c [km/s]); reference library.py:40-72."""
    return 8.97866275, 2.799249247e10, 6371.0, 299_792.458


def den2freq(density):
    """Plasma frequency [Hz] of an electron density [m^-3]; reference library.py:75-97."""
    cp = constants()[0]
    if np.any(np.asarray(density) < 0):
        raise ValueError("Density must be non-negative")
    return np.sqrt(density) * cp


def freq2den(frequency):
    """Electron density [m^-3] of a plasma frequency [Hz]; reference library.py:100-117."""
    cp = constants()[0]
    return (frequency / cp) ** 2


def find_X(n_e, f):
    """X = (f_N / f)^2; reference library.py:120-137."""
    return den2freq(n_e) ** 2 / f ** 2


def find_Y(f, b):
    """Y = f_H / f; reference library.py:140-158."""
    g_p = constants()[1]
    return g_p * b / f


def smooth_nonuniform_grid(start, end, n_points, sharpness):
    """Monotone grid from 0 to 1 that is dense near 1; reference library.py:296-321.

    (As in the reference, ``start``/``end`` only scale the exponential factor; the operator
    calls it with 0, 1, n_points, 10.)
    """
    u = np.linspace(0.0, 1.0, n_points)
    factor = (np.exp(sharpness * (1.0 - u)) - 1.0) / (np.exp(sharpness) - 1.0)
    return 1.0 - (start + (end - start) * factor)


def vertical_to_magnetic_angle(inclination_deg):
    """Angle between the vertical and the field; reference library.py:441-456."""
    return 90.0 - np.abs(inclination_deg)


# ----------------------------------------------------------------------------------------
# the operator
# ----------------------------------------------------------------------------------------
_MODE_CODE = {"O": _native.MODE_O, "X": _native.MODE_X}
_mult_cache = {}


def _default_math(mode_code, math):
    """Default MATH_AUTO: O mode is ill conditioned near reflection and keeps the reference's
    operation order; X mode (conditioning ~1e-11) takes the fast tier.  DESIGN.md section 5."""
    return MATH_AUTO if math is None else int(math)


def find_mu_mup(X, Y, bpsi, mode, *, device=None, math=None):
    """Appleton-Hartree phase and group refractive indices on the GPU.

    Same arguments and results as the reference's ``find_mu_mup`` (library.py:161-256): ``X``,
    ``Y``, ``bpsi`` (degrees) broadcastable arrays, ``mode`` 'O' or 'X'; returns ``(mu, mup)``
    with the inputs' broadcast shape.  The isotropic formulas are used when ``nanmax|Y| <
    1e-12`` over the whole array, as in the reference.  Default tier: faithful.
    """
    if mode not in _MODE_CODE:
        raise ValueError("Mode must be O or X")                   # reference library.py:225-226
    X, Y, bpsi = np.broadcast_arrays(np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64),
                                     np.asarray(bpsi, dtype=np.float64))
    shape = X.shape
    x, y, p = (np.ascontiguousarray(a).reshape(-1) for a in (X, Y, bpsi))
    mu = np.empty(x.size, dtype=np.float64)
    mup = np.empty(x.size, dtype=np.float64)
    ctx = _native.host_context(device)
    ctx.set_math(MATH_FAITHFUL if math is None else int(math))
    _native.raise_for(ctx.mu_mup(x.ctypes.data, y.ctypes.data, p.ctypes.data, x.size, _MODE_CODE[mode],
                                 mu.ctypes.data, mup.ctypes.data, 0))
    return mu.reshape(shape), mup.reshape(shape)


def find_vh(X, Y, bpsi, dh, alt_min, mode, *, device=None, math=None):
    """Virtual height from already-regridded arrays; the reference's ``find_vh`` (library.py:259-293).

    ``X, Y, bpsi, dh``: ``(F, N)`` arrays; returns ``(F,)``: ``nansum(mu' * dh, axis=1)`` with an exact
    zero mapped to NaN, plus ``alt_min``.  Default tier: faithful.
    """
    if mode not in _MODE_CODE:
        raise ValueError("Mode must be O or X")                   # raised by find_mu_mup, library.py:225-226
    X, Y, bpsi, dh = np.broadcast_arrays(*(np.asarray(a, dtype=np.float64) for a in (X, Y, bpsi, dh)))
    if X.ndim != 2:
        raise ValueError("X, Y, bpsi and dh must be 2-D (frequencies x grid points)")
    x, y, p, d = (np.ascontiguousarray(a) for a in (X, Y, bpsi, dh))
    vh = np.empty(x.shape[0], dtype=np.float64)
    ctx = _native.host_context(device)
    ctx.set_math(MATH_FAITHFUL if math is None else int(math))
    _native.raise_for(ctx.find_vh(x.ctypes.data, y.ctypes.data, p.ctypes.data, d.ctypes.data, x.shape[0],
                                  x.shape[1], alt_min, _MODE_CODE[mode], vh.ctypes.data, 0))
    return vh


def regrid_to_nonuniform_grid(f, n_e, b, bpsi, aalt, mode='O', n_points=200, dh=1e-6, *, device=None):
    """Regrid one profile to the stretched per-frequency grid; the reference's function of the same
    name (library.py:324-438), computed on the GPU.

    ``f`` in **Hz**; returns the reference's dict of ``(F, N)`` arrays ``freq, den, bmag, bpsi, dist, alt,
    crit_height`` (float64) and ``ind`` (int64), bit-identical to NumPy's.  As in the reference the
    ``dh`` argument is ignored (it is overwritten with 1e-6 km, library.py:378).
    """
    code = _mode_code(mode)
    fz = np.ascontiguousarray(np.atleast_1d(np.asarray(f)), dtype=np.float64)
    d, bb, p, a = (np.ascontiguousarray(np.asarray(x), dtype=np.float64) for x in (n_e, b, bpsi, aalt))
    if not (d.ndim == 1 and d.shape == bb.shape == p.shape == a.shape):
        raise ValueError("n_e, b, bpsi and aalt must be 1-D arrays of one length")
    mult = _multiplier(n_points)
    shape = (fz.size, int(n_points))
    names = ("freq", "den", "bmag", "bpsi", "dist", "alt", "crit_height")
    out = {k: np.empty(shape, dtype=np.float64) for k in names}
    out["ind"] = np.empty(shape, dtype=np.int64)
    ctx = _native.host_context(device)
    rc = ctx.regrid(fz.ctypes.data, fz.size, d.ctypes.data, bb.ctypes.data, p.ctypes.data, a.ctypes.data, d.size,
                    mult.ctypes.data, int(n_points), code, [out[k].ctypes.data for k in names + ("ind",)], 0)
    _native.raise_for(rc)
    return out


def _mode_code(mode):
    try:
        return _MODE_CODE[mode]
    except (KeyError, TypeError):
        raise ValueError("mode must be 'O' or 'X'") from None     # reference library.py:395-396


def _multiplier(n_points):
    n = int(n_points)
    if n < 1:
        raise ValueError("n_points must be >= 1")
    m = _mult_cache.get(n)
    if m is None:
        with np.errstate(all="ignore"):
            m = np.ascontiguousarray(smooth_nonuniform_grid(0, 1, n, 10.0), dtype=np.float64)
        m.flags.writeable = False
        if len(_mult_cache) < 12:            # entries are never evicted: the library caches their device copies by address
            _mult_cache[n] = m
    return m


def _grid_flag(m, n_points):
    """PRHF_FLAG_GRID_STABLE for a host grid that is one of the cached (never freed, never written) ones."""
    return _native.FLAG_GRID_STABLE if _mult_cache.get(int(n_points)) is m else 0


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch" and hasattr(x, "data_ptr")


def _as_rows(name, x):
    if type(x) is np.ndarray and x.dtype == np.float64 and x.flags.c_contiguous:
        a = x                                        # the usual call: nothing to convert
    else:
        a = np.ascontiguousarray(np.asarray(x), dtype=np.float64)
    if a.ndim == 0 or a.ndim > 2:
        raise ValueError(f"{name} must be 1-D (one profile) or 2-D (profiles x levels)")
    return a


_ptr_cache = {}


def _ptr(x):
    """Address of a C-contiguous float64 array.  ``x.ctypes.data`` builds a ctypes object per access (~1 us): a
    sweep calls the operator with the same ``freq``, ``alt``, ``bmag``, ``bpsi`` arrays again and again, so the address
    is remembered per array object (weak reference: an id that was recycled for another array does not match)."""
    key = id(x)
    hit = _ptr_cache.get(key)
    if hit is not None and hit[0]() is x and hit[2] == x.size:      # (size: a forced in-place resize moves the data)
        return hit[1]
    p = x.ctypes.data
    try:
        if len(_ptr_cache) > 64:
            _ptr_cache.clear()
        _ptr_cache[key] = (weakref.ref(x), p, x.size)
    except TypeError:                                # (an ndarray subclass without weak references)
        pass
    return p


def _np_operator(freq, den, bmag, bpsi, alt, mode_code, n_points, device, math, devices=None):
    if type(freq) is np.ndarray and freq.ndim == 1 and freq.dtype == np.float64 and freq.flags.c_contiguous:
        f = freq
    else:
        f = np.ascontiguousarray(np.atleast_1d(np.asarray(freq)), dtype=np.float64)
    if f.ndim != 1:
        raise ValueError("freq must be a scalar or 1-D")
    d, b, p, a = _as_rows("den", den), _as_rows("bmag", bmag), _as_rows("bpsi", bpsi), _as_rows("alt", alt)
    single = d.ndim == 1
    # batch extension: one bmag / bpsi row may serve every density row (an ensemble or a fit at one site)
    shared = d.ndim == 2 and b.ndim == 1 and p.ndim == 1 and b.shape == p.shape == d.shape[1:]
    d2 = d.reshape(1, -1) if single else d
    b2, p2 = (b, p) if shared else (b.reshape(1, -1) if b.ndim == 1 else b, p.reshape(1, -1) if p.ndim == 1 else p)
    if not shared and not (d2.shape == b2.shape == p2.shape):
        logger.error("Error: freq, den, bmag, bpsi, alt should have same size")   # reference library.py:487-488
        raise ValueError("den, bmag and bpsi must have the same shape")
    n_prof, n_alt = d2.shape
    if a.shape[-1] != n_alt or (a.ndim == 2 and a.shape[0] != n_prof):
        logger.error("Error: freq, den, bmag, bpsi, alt should have same size")
        raise ValueError("alt must have one value per density level")
    alt_stride = n_alt if a.ndim == 2 else 0
    mult = _multiplier(n_points)
    out = np.empty((n_prof, f.size), dtype=np.float64)
    flags = (_native.FLAG_SHARED_FIELD if shared else 0) | _grid_flag(mult, n_points)
    level = _default_math(mode_code, math)
    if devices is not None:
        ids = _device_list(devices)
        if len(ids) > 1 and n_prof >= 2 * len(ids):
            _run_on_devices(ids, f, d2, b2, p2, a, shared, alt_stride, mult, int(n_points), mode_code, level, flags, out)
            return out
        device = ids[0]
    ctx = _native.host_context(device)
    ctx.set_math(level)
    # (d, b, p: the caller's own array objects - their reshaped views share the address)
    rc = ctx.vfo_batch(_ptr(f), f.size, _ptr(d), _ptr(b), _ptr(p), _ptr(a),
                       n_prof, n_alt, n_alt, alt_stride, _ptr(mult), int(n_points), mode_code,
                       out.ctypes.data, flags)
    _native.raise_for(rc)
    return out[0] if single else out


def _device_list(devices):
    """``"all"`` -> every visible GPU; else a sequence of device indices (an index may repeat: two contexts on it)."""
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError("devices is 'all' or a sequence of GPU indices")
        return list(range(_native.device_count()))
    ids = [int(x) for x in devices]
    if not ids:
        raise ValueError("devices is empty")
    return ids


class _DeviceWorker:
    """One host thread that owns one context on one GPU (the library's contexts are per thread and device, and ctypes
    releases the GIL during a call): the drop-in call's way to the other GPUs of the node without a process group."""

    def __init__(self, device):
        import queue
        import threading
        self.device = device
        self.jobs = queue.SimpleQueue()
        self.thread = threading.Thread(target=self._loop, name=f"prhf-device-{device}", daemon=True)
        self.thread.start()

    def _loop(self):
        while True:
            fn, done = self.jobs.get()
            try:
                done.append((True, fn(_native.host_context(self.device))))
            except BaseException as exc:                # noqa: BLE001 - handed to the caller's thread
                done.append((False, exc))
            finally:
                done_event = done[0]
                done_event.set()

    def submit(self, fn):
        import threading
        done = [threading.Event()]
        self.jobs.put((fn, done))
        return done


_device_workers = {}


def _run_on_devices(ids, f, d2, b2, p2, a, shared, alt_stride, mult, n_points, mode_code, level, flags, out):
    """Rows cut into contiguous blocks (``dist.shard_bounds``), one block per entry of ``ids``, each evaluated by that
    entry's own thread and context straight into its rows of ``out`` - no process group, no collective: the result
    rows ARE the gather.  Bit-identical to the single-device call (a launch's rows do not depend on their neighbours)."""
    from .dist import shard_bounds
    n_prof, n_alt = d2.shape
    n_freq = f.size
    pending = []
    for slot, dev in enumerate(ids):
        lo, hi = shard_bounds(n_prof, len(ids), slot)
        if hi <= lo:
            continue
        worker = _device_workers.get((slot, dev))
        if worker is None:
            worker = _device_workers[(slot, dev)] = _DeviceWorker(dev)

        def job(ctx, lo=lo, hi=hi):
            ctx.set_math(level)
            row = lo * n_alt * 8
            return ctx.vfo_batch(f.ctypes.data, n_freq, d2.ctypes.data + row,
                                 b2.ctypes.data + (0 if shared else row), p2.ctypes.data + (0 if shared else row),
                                 a.ctypes.data + (row if alt_stride else 0), hi - lo, n_alt, n_alt, alt_stride,
                                 mult.ctypes.data, n_points, mode_code, out.ctypes.data + lo * n_freq * 8, flags), _native.last_error()
        pending.append(worker.submit(job))
    failure = None
    for done in pending:
        done[0].wait()
        ok, value = done[1]
        if not ok:
            failure = failure or value
        elif value[0] != _native.OK and failure is None:
            failure = _native.error_for(*value)        # (prhf_last_error is per thread: the worker read its own message)
    if failure is not None:
        raise failure


def _torch_operator(freq, den, bmag, bpsi, alt, mode_code, n_points, math, sync, out):
    import torch

    dev = den.device
    if dev.type != "cuda":
        raise ValueError("torch inputs must live on the GPU; pass NumPy arrays for host data")

    def prep(x, name):
        if not _is_torch(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float64), device=dev)
        if x.device != dev:
            raise ValueError(f"{name} is on {x.device}, expected {dev}")
        return x.to(torch.float64).contiguous()

    f = prep(freq, "freq").reshape(-1)
    d, b, p, a = (prep(x, n) for x, n in ((den, "den"), (bmag, "bmag"), (bpsi, "bpsi"), (alt, "alt")))
    single = d.dim() == 1
    # (2-D den with 1-D bmag and bpsi: one field row shared by every profile)
    shared = d.dim() == 2 and b.dim() == 1 and p.dim() == 1 and b.shape == p.shape == d.shape[1:]
    d2 = d.reshape(1, -1) if single else d
    b2, p2 = (b, p) if shared else tuple(x.reshape(1, -1) if x.dim() == 1 else x for x in (b, p))
    if d2.dim() != 2 or (not shared and not (d2.shape == b2.shape == p2.shape)):
        raise ValueError("den, bmag and bpsi must have the same 1-D or 2-D shape")
    n_prof, n_alt = d2.shape
    if a.shape[-1] != n_alt or (a.dim() == 2 and a.shape[0] != n_prof):
        raise ValueError("alt must have one value per density level")
    alt_stride = n_alt if a.dim() == 2 else 0
    mult, grid_flag = _device_grid((int(n_points),), dev)
    if out is None:
        out = torch.empty((n_prof, f.numel()), dtype=torch.float64, device=dev)
    elif out.shape != (n_prof, f.numel()) or out.dtype != torch.float64 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float64 tensor of shape (P, F)")
    ctx = _native.context(dev.index if dev.index is not None else torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_math(_default_math(mode_code, math))
    rc = ctx.vfo_batch(f.data_ptr(), f.numel(), d2.data_ptr(), b2.data_ptr(), p2.data_ptr(), a.data_ptr(),
                       n_prof, n_alt, n_alt, alt_stride, mult.data_ptr(), int(n_points), mode_code,
                       out.data_ptr(), _native.FLAG_DEVICE_PTRS | _native.FLAG_ASYNC | grid_flag |
                       (_native.FLAG_SHARED_FIELD if shared else 0))
    _native.raise_for(rc)
    if sync:
        _native.raise_for(ctx.sync())
    return out[0] if single else out


_torch_mult = {}


def _device_grid(n_points_list, dev):
    """Concatenated stretched grids as a device tensor, and the flag that tells the library whether
    the tensor is one of the cached (never freed, never written) ones whose derived tables it may keep."""
    import torch

    key = (tuple(n_points_list), dev.index)
    mult = _torch_mult.get(key)
    if mult is not None:
        return mult, _native.FLAG_GRID_STABLE
    mult = torch.as_tensor(np.concatenate([_multiplier(n) for n in n_points_list]), device=dev)
    if len(_torch_mult) < 64:            # entries are never evicted: a freed address could come back with other contents
        _torch_mult[key] = mult
        return mult, _native.FLAG_GRID_STABLE
    return mult, 0


def vertical_forward_operator(freq, den, bmag, bpsi, alt, mode='O', n_points=200, *,
                              device=None, devices=None, math=None, sync=True, out=None):
    """Virtual height [km] of each sounder frequency for one profile (or a batch).

    Parameters are the reference's (library.py:459-484): ``freq`` MHz, ``den`` m^-3,
    ``bmag`` Tesla, ``bpsi`` degrees, ``alt`` km (ascending), ``mode`` 'O' or 'X',
    ``n_points`` stretched-grid points per frequency.  Returns ``ndarray (F,)`` for 1-D
    profiles (a scalar ``freq`` gives shape (1,)), ``(P, F)`` for 2-D ones; NaN where the
    frequency is not reflected below the density peak.

    Keyword-only extensions: ``device`` (GPU index for host inputs; default ``PRHF_DEVICE``
    / ``LOCAL_RANK`` / 0), ``devices`` (host inputs, 2-D batches: ``"all"`` or a sequence of GPU
    indices - the rows are cut into contiguous blocks, one per entry, each evaluated by its own
    host thread and context straight into its rows of the result; no process group is needed, and
    the values are those of the single-GPU call bit for bit), ``math`` (``MATH_FAITHFUL``: the reference's operation order at every
    grid point; ``MATH_FAST``: the reduced algebra at every point; default ``MATH_AUTO``: fast for
    'X'; for 'O' the reference's order where 1 - X <= 1e-5 - where it decides the answer - and the
    reduced algebra elsewhere, which reproduces the reference to 1e-10),
    and for GPU-resident torch inputs ``sync`` (wait and surface data errors) and ``out``.
    With 2-D ``den``, 1-D ``bmag`` and ``bpsi`` are one field row shared by every profile.

    Raises ``ValueError("mode must be 'O' or 'X'")``, ``ValueError("Density must be
    non-negative")`` (reference library.py:395-396, :93-94), ``IndexError`` when the density
    peak is the first level (the reference fails with IndexError there too), and
    ``ValueError`` on shape mismatch (the reference only logs, library.py:487-488).

    NaN inputs behave as in the reference (fixture G13): a density column padded with NaN is cut at
    the first NaN (``np.argmax``, library.py:371); a NaN in ``alt`` makes the profile's whole trace NaN
    (``np.min(alt)``, :507), and so does a NaN in ``bmag`` below the peak in X mode (:389); in O mode -
    and for a NaN in ``bpsi`` in either mode - the grid points of the two segments next to that level
    drop out of the sum (:288).  A frequency that is not a positive finite number gives NaN for that
    frequency and leaves the others alone (the reference: NaN for 0 and NaN, a meaningless number for a
    negative frequency).  Profiles may have up to 65 535 levels; where the levels BELOW the density
    peak number more than 1400 they no longer fit the GPU's local memory and are staged in global
    memory instead (1.1 - 1.4 times the time per grid point on long grids).
    """
    code = _mode_code(mode)
    if any(_is_torch(x) and x.is_cuda for x in (den, bmag, bpsi)):
        return _torch_operator(freq, den, bmag, bpsi, alt, code, n_points, math, sync, out)
    return _np_operator(freq, den, bmag, bpsi, alt, code, n_points, device, math, devices)


def vertical_forward_operator_mixed(freq, den, bmag, bpsi, alt, segments, *, device=None, math=None, sync=True):
    """Several (profile range, mode, n_points) slices in ONE launch (BASELINE config 5).

    ``segments`` is a sequence of ``(prof_begin, prof_end, mode, n_points)``; profile ranges
    index the rows of the 2-D inputs.  Returns ``(P, F)`` float64; rows not covered by any segment
    are NaN.  GPU-resident torch tensors are used in place (zero copy, launched on torch's current
    stream, synchronised before returning unless ``sync=False``) and give a tensor on the same device.
    """
    if any(_is_torch(x) and x.is_cuda for x in (den, bmag, bpsi)):
        return _torch_mixed(freq, den, bmag, bpsi, alt, segments, math, sync)
    f = np.ascontiguousarray(np.atleast_1d(np.asarray(freq)), dtype=np.float64)
    d2, b2, p2 = (np.atleast_2d(_as_rows(n, x)) for n, x in (("den", den), ("bmag", bmag), ("bpsi", bpsi)))
    a = _as_rows("alt", alt)
    if not (d2.shape == b2.shape == p2.shape):
        raise ValueError("den, bmag and bpsi must have the same shape")
    n_prof, n_alt = d2.shape
    if a.shape[-1] != n_alt or (a.ndim == 2 and a.shape[0] != n_prof):
        raise ValueError("alt must have one value per density level")
    segs, grids, off = [], [], 0
    for (p0, p1, mode, n_points) in segments:
        m = _multiplier(n_points)
        segs.append(_native.Segment(int(p0), int(p1), _mode_code(mode), int(n_points), off, int(p0) * f.size))
        grids.append(m)
        off += m.size
    mult = np.ascontiguousarray(np.concatenate(grids)) if grids else np.zeros(1)
    out = np.full((n_prof, f.size), np.nan, dtype=np.float64)
    if not segs:
        return out
    ctx = _native.host_context(device)
    ctx.set_math(_default_math(None, math))       # MATH_AUTO: each slice in its mode's tier, one launch
    # the library writes only the rows its segments cover: stage `out` through the call
    rc = ctx.vfo_worklist(f.ctypes.data, f.size, d2.ctypes.data, b2.ctypes.data, p2.ctypes.data, a.ctypes.data,
                          n_prof, n_alt, n_alt, n_alt if a.ndim == 2 else 0, mult.ctypes.data, mult.size,
                          segs, out.ctypes.data, 0)
    _native.raise_for(rc)
    covered = np.zeros(n_prof, dtype=bool)
    for s in segs:
        covered[s.prof_begin:s.prof_end] = True
    out[~covered] = np.nan
    return out


def _torch_mixed(freq, den, bmag, bpsi, alt, segments, math, sync=True):
    import torch

    dev = den.device

    def prep(x):
        if not _is_torch(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float64), device=dev)
        return x.to(device=dev, dtype=torch.float64).contiguous()

    f = prep(freq).reshape(-1)
    d2, b2, p2, a = prep(den), prep(bmag), prep(bpsi), prep(alt)
    if d2.dim() != 2 or not (d2.shape == b2.shape == p2.shape):
        raise ValueError("den, bmag and bpsi must share a (P, N_alt) shape")
    n_prof, n_alt = d2.shape
    if a.shape[-1] != n_alt or (a.dim() == 2 and a.shape[0] != n_prof):
        raise ValueError("alt must have one value per density level")
    segs, sizes, off = [], [], 0
    for (p0, p1, mode, n_points) in segments:
        n = _multiplier(n_points).size
        segs.append(_native.Segment(int(p0), int(p1), _mode_code(mode), int(n_points), off, int(p0) * f.numel()))
        sizes.append(n)
        off += n
    out = torch.full((n_prof, f.numel()), float("nan"), dtype=torch.float64, device=dev)
    if not segs:
        return out
    mult, grid_flag = _device_grid(sizes, dev)
    ctx = _native.context(dev.index if dev.index is not None else torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_math(_default_math(None, math))
    rc = ctx.vfo_worklist(f.data_ptr(), f.numel(), d2.data_ptr(), b2.data_ptr(), p2.data_ptr(), a.data_ptr(),
                          n_prof, n_alt, n_alt, n_alt if a.dim() == 2 else 0, mult.data_ptr(), mult.numel(), segs,
                          out.data_ptr(), _native.FLAG_DEVICE_PTRS | _native.FLAG_ASYNC | grid_flag)
    _native.raise_for(rc)
    if sync or not grid_flag:
        _native.raise_for(ctx.sync())          # an uncached `mult` must stay alive until the kernel has read it
    return out


def set_option(name, value, device=None):
    """A launch-shaping or arithmetic setting of this thread's context on ``device`` (``prhf_ctx_set_option``
    in include/prhf.h: tests and A/B measurements; the defaults are the measured best).  The library reads
    no environment variable."""
    _native.context(device).set_option(name, value)


def last_kernel_ms(device=None):
    """Device time [ms] of the most recent TIMED launch on this thread's context: every launch on GPU-resident
    inputs, and calls on NumPy arrays after ``set_option("timing", 1)`` (off by default: the two event records are
    3.5 us of a 41 us single-profile call)."""
    return _native.context(device).last_kernel_ms()


def recent_kernel_ms(count=64, device=None):
    """Device times [ms] of the most recent launches on this thread's context (at most 64), oldest first."""
    return _native.context(device).recent_kernel_ms(count)
