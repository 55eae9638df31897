"""ctypes wrapper of oracle/libvfo_oracle.so, the plain-C restatement (TEST INFRASTRUCTURE ONLY).

Same rules as oracle/vfo_numpy.py: only tests/, smoke() and bench.py's cpu_baseline leg use it.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

from . import vfo_numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvfo_oracle.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def require():
    """The checker must not go missing silently: build the library from oracle/vfo_oracle.c if it is absent (gcc, as
    __graft_entry__.build() does) and raise when that fails.  Tests call this instead of skipping."""
    if not available():
        import subprocess
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    if not available():
        raise RuntimeError("oracle/libvfo_oracle.so is missing and could not be built (make -C oracle)")
    return True


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIB_PATH)
        lib.vfo_oracle_batch.restype = ctypes.c_int
        lib.vfo_oracle_batch.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4 + \
            [ctypes.c_int64] * 3 + [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        lib.vfo_oracle_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def threads():
    return _load().vfo_oracle_threads()


def virtual_heights_batch(freq_mhz, den, bmag, bpsi, alt, mode="O", n_points=200, n_threads=0):
    """(P, N_alt) -> (P, F); 1-D profiles give (F,).  OpenMP over profiles (0 = all cores)."""
    if mode not in ("O", "X"):
        raise ValueError("mode must be 'O' or 'X'")
    f = np.ascontiguousarray(np.atleast_1d(freq_mhz), dtype=np.float64)
    single = np.ndim(den) == 1
    d, b, p = (np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64) for x in (den, bmag, bpsi))
    a = np.ascontiguousarray(alt, dtype=np.float64)
    mult = np.ascontiguousarray(vfo_numpy.stretch_multiplier(n_points))
    out = np.empty((d.shape[0], f.size))
    rc = _load().vfo_oracle_batch(f.ctypes.data, f.size, d.ctypes.data, b.ctypes.data, p.ctypes.data, a.ctypes.data,
                                  d.shape[0], d.shape[1], d.shape[1] if a.ndim == 2 else 0, mult.ctypes.data,
                                  int(n_points), 0 if mode == "O" else 1, out.ctypes.data, int(n_threads))
    if rc == -2:
        raise ValueError("Density must be non-negative")
    if rc == -3:
        raise IndexError("density peak at index 0")
    return out[0] if single else out
