"""ctypes binding of libprhf.so (C ABI: include/prhf.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``make -C pyrayhf_amd/csrc`` and is the only compute path: if it is missing, or no GPU is
visible, the operator raises - there is no CPU fallback in this package.
"""

from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRHF_LIB") or os.path.join(_HERE, "libprhf.so")   # PRHF_LIB: A/B builds

OK, EINVAL, ENEGDEN, EPEAK0, EHIP, ENOMEM = 0, -1, -2, -3, -4, -5
MODE_O, MODE_X = 0, 1
FLAG_DEVICE_PTRS, FLAG_ASYNC, FLAG_GRID_STABLE, FLAG_SHARED_FIELD = 0x1, 0x2, 0x4, 0x8
MATH_FAITHFUL, MATH_FAST, MATH_AUTO = 0, 1, 2
ABI_VERSION = 3

c_double_p = ctypes.POINTER(ctypes.c_double)


class Segment(ctypes.Structure):
    """``prhf_segment`` of include/prhf.h."""
    _fields_ = [("prof_begin", ctypes.c_int64), ("prof_end", ctypes.c_int64),
                ("mode", ctypes.c_int32), ("n_points", ctypes.c_int32),
                ("mult_offset", ctypes.c_int64), ("out_offset", ctypes.c_int64)]


class NativeLibraryError(RuntimeError):
    """libprhf.so is missing, stale, or the HIP runtime failed."""


# every symbol include/prhf.h declares: name -> (restype, argtypes)
_PROTOTYPES = {
    "prhf_abi_version": (ctypes.c_int, []),
    "prhf_last_error": (ctypes.c_char_p, []),
    "prhf_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "prhf_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "prhf_ctx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "prhf_ctx_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "prhf_ctx_set_math": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "prhf_ctx_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]),
    "prhf_vfo_batch_f64": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]),
    "prhf_vfo_worklist_f64": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_int64, ctypes.POINTER(Segment), ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]),
    "prhf_mu_mup_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_uint32]),
    "prhf_find_vh_f64": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_void_p] * 4 +
                         [ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_int32, ctypes.c_void_p,
                          ctypes.c_uint32]),
    "prhf_regrid_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4 +
                        [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 8 +
                        [ctypes.c_uint32]),
    "prhf_residual_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]),
    "prhf_vfo_residual_f64": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_uint32]),
    "prhf_snell_cartesian_f64": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32]),
    "prhf_snell_spherical_f64": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32]),
    "prhf_snell_fan_f64": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, ctypes.c_double,
        ctypes.c_double, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_uint32]),
    "prhf_occupancy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                      ctypes.POINTER(ctypes.c_int32)]),
    "prhf_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "prhf_last_kernel_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "prhf_recent_kernel_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int32,
                                             ctypes.POINTER(ctypes.c_int32)]),
}

_lib = None
_lib_lock = threading.Lock()


def load():
    """Load libprhf.so once and attach prototypes; raise NativeLibraryError if unusable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pyrayhf_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own HIP runtime.  If libprhf.so pulled in the system runtime first, a
        # later `import torch` would bring a second runtime into the process and that one finds no GPU
        # ("No HIP GPUs are available").  Loading torch first makes both share torch's runtime.
        if os.environ.get("PRHF_NO_TORCH_PRELOAD", "") == "":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as exc:
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
        for name, (res, args) in _PROTOTYPES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
            fn.restype = res
            fn.argtypes = args
        if lib.prhf_abi_version() != ABI_VERSION:
            raise NativeLibraryError(f"ABI version {lib.prhf_abi_version()} != {ABI_VERSION}; rebuild libprhf.so")
        _lib = lib
        return _lib


def exported_symbols():
    return sorted(_PROTOTYPES)


def last_error():
    msg = load().prhf_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def error_for(code, msg):
    """The exception the reference raises for the condition a PRHF_E* code stands for (``msg``: prhf_last_error of
    the thread that made the call)."""
    if code == ENEGDEN:
        return ValueError("Density must be non-negative")          # reference library.py:93-94
    if code == EPEAK0:
        return IndexError(msg or "density peak at index 0")         # the reference fails with IndexError too
    if code == EINVAL:
        return ValueError(msg or "invalid argument")
    if code == ENOMEM:
        return MemoryError(msg)
    return NativeLibraryError(msg or f"libprhf error {code}")


def raise_for(code):
    """Map a PRHF_E* code to the exception the reference raises for the same condition."""
    if code == OK:
        return
    raise error_for(code, last_error())


def device_count():
    n = ctypes.c_int(0)
    raise_for(load().prhf_device_count(ctypes.byref(n)))
    return n.value


class Context:
    """Owns one ``prhf_ctx`` (one device, one stream, scratch buffers)."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        raise_for(self._lib.prhf_ctx_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)
        # what the library was last told, so that the per-call setters cost a comparison, not a foreign call
        self._math = None
        self._stream = (0, False)          # (handle, borrowed): the context starts on its own stream

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.prhf_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass

    def set_stream(self, stream_ptr, borrow=True):
        """Launch on the caller's stream (``stream_ptr`` 0 = the legacy default stream) or, with
        ``borrow=False``, on the context's own stream."""
        want = (int(stream_ptr or 0), bool(borrow)) if borrow else (0, False)
        if want == self._stream:
            return
        raise_for(self._lib.prhf_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr or None), 1 if borrow else 0))
        self._stream = want

    def set_math(self, level):
        level = int(level)
        if level == self._math:
            return
        raise_for(self._lib.prhf_ctx_set_math(self._h, level))
        self._math = level

    def set_option(self, name, value):
        """A launch-shaping or arithmetic setting of this context (include/prhf.h, prhf_ctx_set_option)."""
        raise_for(self._lib.prhf_ctx_set_option(self._h, str(name).encode(), float(value)))

    def vfo_batch(self, freq, n_freq, den, bmag, bpsi, alt, n_prof, n_alt, prof_stride, alt_stride,
                  mult, n_points, mode, out, flags):
        """All array arguments are raw addresses (ints)."""
        return self._lib.prhf_vfo_batch_f64(self._h, freq, n_freq, den, bmag, bpsi, alt, n_prof, n_alt,
                                            prof_stride, alt_stride, mult, n_points, mode, out, flags)

    def vfo_worklist(self, freq, n_freq, den, bmag, bpsi, alt, n_prof, n_alt, prof_stride, alt_stride,
                     mult, mult_len, segments, out, flags):
        arr = (Segment * len(segments))(*segments)
        return self._lib.prhf_vfo_worklist_f64(self._h, freq, n_freq, den, bmag, bpsi, alt, n_prof, n_alt,
                                               prof_stride, alt_stride, mult, mult_len, arr, len(segments),
                                               out, flags)

    def mu_mup(self, X, Y, psi, n, mode, mu, mup, flags):
        return self._lib.prhf_mu_mup_f64(self._h, X, Y, psi, n, mode, mu, mup, flags)

    def find_vh(self, X, Y, psi, dh, n_rows, n_cols, alt_min, mode, vh, flags):
        return self._lib.prhf_find_vh_f64(self._h, X, Y, psi, dh, n_rows, n_cols, float(alt_min), mode, vh, flags)

    def regrid(self, freq_hz, n_freq, den, bmag, bpsi, alt, n_alt, mult, n_points, mode, outs, flags):
        """outs: eight raw addresses in the order freq, den, bmag, bpsi, dist, alt, crit_height, ind."""
        return self._lib.prhf_regrid_f64(self._h, freq_hz, n_freq, den, bmag, bpsi, alt, n_alt, mult, n_points,
                                         mode, *outs, flags)

    def residual(self, vh_model, vh_obs, n_prof, n_freq, residual, cost, flags):
        return self._lib.prhf_residual_f64(self._h, vh_model, vh_obs, n_prof, n_freq, residual or None,
                                           cost or None, flags)

    def vfo_residual(self, freq, n_freq, den, bmag, bpsi, alt, n_prof, n_alt, prof_stride, alt_stride, mult,
                     n_points, mode, vh_obs, vh, residual, cost, flags):
        return self._lib.prhf_vfo_residual_f64(self._h, freq, n_freq, den, bmag, bpsi, alt, n_prof, n_alt,
                                               prof_stride, alt_stride, mult, n_points, mode, vh_obs,
                                               vh or None, residual or None, cost or None, flags)

    def snell_cartesian(self, freq_hz, elev, prof_idx, n_rays, den, bmag, bpsi, alt, n_prof, n_alt, alt_stride,
                        mode, out, path_x, path_z, path_stride, flags):
        return self._lib.prhf_snell_cartesian_f64(self._h, freq_hz, elev, prof_idx or None, n_rays, den, bmag, bpsi,
                                                  alt, n_prof, n_alt, alt_stride, mode, out, path_x or None,
                                                  path_z or None, path_stride, flags)

    def snell_spherical(self, freq_hz, elev, prof_idx, n_rays, den, bmag, bpsi, alt, n_prof, n_alt, alt_stride,
                        mode, r_e, dz_target, apex_boost, max_substeps, out, path_x, path_z, path_stride, flags):
        return self._lib.prhf_snell_spherical_f64(self._h, freq_hz, elev, prof_idx or None, n_rays, den, bmag, bpsi,
                                                  alt, n_prof, n_alt, alt_stride, mode, float(r_e), float(dz_target),
                                                  float(apex_boost), int(max_substeps), out, path_x or None,
                                                  path_z or None, path_stride, flags)

    def snell_fan(self, geometry, group_freq, group_prof, n_groups, ray_group, elev, n_rays, den, bmag, bpsi, alt, n_prof,
                  n_alt, alt_stride, mode, r_e, dz_target, apex_boost, max_substeps, out, path_x, path_z, path_stride,
                  flags):
        return self._lib.prhf_snell_fan_f64(self._h, int(geometry), group_freq, group_prof or None, n_groups, ray_group,
                                            elev, n_rays, den, bmag, bpsi, alt, n_prof, n_alt, alt_stride, mode,
                                            float(r_e), float(dz_target), float(apex_boost), int(max_substeps), out,
                                            path_x or None, path_z or None, path_stride, flags)

    def occupancy(self, n_alt, math):
        n = ctypes.c_int32(0)
        raise_for(self._lib.prhf_occupancy(self._h, int(n_alt), int(math), ctypes.byref(n)))
        return n.value

    def sync(self):
        return self._lib.prhf_sync(self._h)

    def last_kernel_ms(self):
        ms = ctypes.c_double(0.0)
        raise_for(self._lib.prhf_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def recent_kernel_ms(self, count=64):
        """Device times [ms] of the most recent launches (at most 64 are remembered), oldest first; one
        synchronisation on the newest."""
        buf = (ctypes.c_double * max(int(count), 1))()
        n = ctypes.c_int32(0)
        raise_for(self._lib.prhf_recent_kernel_ms(self._h, buf, int(count), ctypes.byref(n)))
        return [buf[i] for i in range(n.value)]


_tls = threading.local()


def default_device():
    for key in ("PRHF_DEVICE", "LOCAL_RANK"):
        if os.environ.get(key, "") != "":
            return int(os.environ[key])
    return 0


def host_context(device=None):
    """The cached context, launching on its OWN stream: what every host-buffer (NumPy) call uses.  A
    previous torch call on this thread may have left the context on torch's stream (borrowed); a host
    call must not keep running there - that stream may be the legacy blocking one, or already destroyed."""
    ctx = context(device)
    ctx.set_stream(0, borrow=False)
    return ctx


def context(device=None):
    """Per-thread, per-device cached context."""
    dev = default_device() if device is None else int(device)
    cache = getattr(_tls, "ctx", None)
    if cache is None:
        cache = _tls.ctx = {}
    ctx = cache.get(dev)
    if ctx is None:
        ctx = cache[dev] = Context(dev)
    return ctx
