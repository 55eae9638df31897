"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/prhf.h declares (no compute calls: there is no GPU here), and the host logic of the
drop-in (argument checking, helper functions) behaves like the reference."""

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_golden
from pyrayhf_amd import _native, library


def _header_symbols():
    text = open(os.path.join(REPO, "include", "prhf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prhf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _header_symbols() == _native.exported_symbols()


def test_library_loads_and_exports_every_symbol():
    lib = _native.load()
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.prhf_abi_version() == _native.ABI_VERSION
    assert ctypes.sizeof(_native.Segment) == 40        # prhf_segment layout


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    g = load_golden("g1_basic.npz")
    with pytest.raises((_native.NativeLibraryError, ValueError)):
        library.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 50)


def test_mode_is_validated_before_anything_else():
    g = load_golden("g1_basic.npz")
    with pytest.raises(ValueError, match="mode must be 'O' or 'X'"):       # reference library.py:395-396
        library.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "Z", 50)


def test_host_helpers_match_reference_known_answers():
    g = load_golden("g3_index_kat.npz")
    assert np.array_equal(np.array(library.constants()), g["constants"])          # reference test_core.py:38-44
    assert np.array_equal(library.smooth_nonuniform_grid(0, 1, 10, 10.0), g["grid10"])
    grid = library.smooth_nonuniform_grid(0.0, 1.0, 10, 5.0)                        # reference test_core.py:171-188
    assert len(grid) == 10 and np.all(np.diff(grid) > 0)
    assert np.isclose(grid[0], 0.0, atol=1e-6) and np.isclose(grid[-1], 1.0, atol=1e-6)
    den = np.array([1.0e12, 2.5e12, 0.0])
    assert np.allclose(library.den2freq(den), np.sqrt(den) * 8.97866275, rtol=1e-8)   # test_core.py:57-65
    assert isinstance(library.den2freq(1.0e12), float)
    f = np.array([8.97866275e6, 2 * 8.97866275e6, 0.0])
    assert np.allclose(library.freq2den(f), (f / 8.97866275) ** 2, rtol=1e-8)          # test_core.py:78-86
    n_e, fr = np.array([1.0e12, 2.5e12, 0.0]), np.array([1.0e7, 1.5e7, 2.0e7])
    assert np.allclose(library.find_X(n_e, fr), (np.sqrt(n_e) * 8.97866275) ** 2 / fr ** 2, rtol=1e-8)
    b = np.array([5.0e-5, 6.0e-5, 7.0e-5])
    assert np.allclose(library.find_Y(fr, b), 2.799249247e10 * b / fr, rtol=1e-8)      # test_core.py:124-134
    with pytest.raises(ValueError, match="Density must be non-negative"):
        library.den2freq(np.array([-1.0]))
    assert library.vertical_to_magnetic_angle(60.0) == 30.0                             # test_core.py:210-220
    assert np.allclose(library.vertical_to_magnetic_angle(np.array([0.0, 45.0, 90.0])), [90.0, 45.0, 0.0])


def test_argument_errors_are_reported_without_a_gpu():
    """Entry points validate their arguments before touching the device: a null context is PRHF_EINVAL
    with a thread-local message, and creating a context without a GPU is PRHF_EHIP (or EINVAL: no such
    device) - never a silent success."""
    lib = _native.load()
    rc = lib.prhf_vfo_batch_f64(None, None, 0, None, None, None, None, 0, 0, 0, 0, None, 0, 0, None, 0)
    assert rc == _native.EINVAL and b"context" in lib.prhf_last_error()
    assert lib.prhf_sync(None) == _native.EINVAL
    assert lib.prhf_mu_mup_f64(None, None, None, None, 0, 0, None, None, 0) == _native.EINVAL
    assert lib.prhf_ctx_set_math(None, 0) == _native.EINVAL
    import torch
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        rc = lib.prhf_ctx_create(0, ctypes.byref(h))
        assert rc in (_native.EHIP, _native.EINVAL) and not h.value
        assert lib.prhf_last_error()
    assert lib.prhf_ctx_destroy(None) == _native.OK


def test_shape_and_argument_validation_happens_on_the_host():
    """Bad shapes are rejected before any device is touched (so also without a GPU)."""
    g = load_golden("g1_basic.npz")
    with pytest.raises(ValueError):
        library.vertical_forward_operator(g["freq"], g["den"][:2], g["bmag"], g["bpsi"], g["alt"], "O", 10)
    with pytest.raises(ValueError):
        library.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"][:2], "O", 10)
    with pytest.raises(ValueError):
        library.vertical_forward_operator(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], "O", 0)
    with pytest.raises(ValueError):
        library.vertical_forward_operator(g["freq"].reshape(1, -1, 1), g["den"], g["bmag"], g["bpsi"], g["alt"])
    with pytest.raises(ValueError, match="Mode must be O or X"):
        library.find_mu_mup(np.ones(3), np.ones(3), np.ones(3), "Q")
    with pytest.raises(ValueError, match="mode must be 'O' or 'X'"):
        library.regrid_to_nonuniform_grid(g["freq"], g["den"], g["bmag"], g["bpsi"], g["alt"], mode="Q")


def test_missing_library_is_a_loud_error(tmp_path, monkeypatch):
    """A missing libprhf.so must raise, never fall back (checked in a fresh interpreter)."""
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from pyrayhf_amd import library, _native\n"
            "try:\n"
            "    library.vertical_forward_operator(np.array([1.0, 2.0]), np.array([0, 5e11, 1e12]), np.full(3, 5e-5),\n"
            "                                      np.full(3, 60.0), np.array([100.0, 200.0, 300.0]))\n"
            "except _native.NativeLibraryError as exc:\n"
            "    print('LOUD', 'not found' in str(exc))\n" % REPO)
    env = dict(os.environ, PRHF_LIB=str(tmp_path / "nowhere" / "libprhf.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert "LOUD True" in out.stdout, out.stdout + out.stderr


def test_header_constants_match_the_binding():
    """#define values of include/prhf.h (codes, flags, modes, arithmetic settings) against pyrayhf_amd/_native.py."""
    import re
    from pyrayhf_amd import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "prhf.h")).read()
    defines = {m.group(1): int(m.group(2).rstrip("u"), 0)
               for m in re.finditer(r"#define\s+(PRHF_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+)u?)\b", text)}
    expect = {"PRHF_OK": _native.OK, "PRHF_EINVAL": _native.EINVAL, "PRHF_ENEGDEN": _native.ENEGDEN,
              "PRHF_EPEAK0": _native.EPEAK0, "PRHF_EHIP": _native.EHIP, "PRHF_ENOMEM": _native.ENOMEM,
              "PRHF_FLAG_DEVICE_PTRS": _native.FLAG_DEVICE_PTRS, "PRHF_FLAG_ASYNC": _native.FLAG_ASYNC,
              "PRHF_FLAG_GRID_STABLE": _native.FLAG_GRID_STABLE, "PRHF_FLAG_SHARED_FIELD": _native.FLAG_SHARED_FIELD,
              "PRHF_MODE_O": _native.MODE_O,
              "PRHF_MODE_X": _native.MODE_X, "PRHF_MATH_FAITHFUL": _native.MATH_FAITHFUL,
              "PRHF_MATH_FAST": _native.MATH_FAST, "PRHF_MATH_AUTO": _native.MATH_AUTO,
              "PRHF_ABI_VERSION": _native.ABI_VERSION}
    for name, value in expect.items():
        assert defines.get(name) == value, (name, defines.get(name), value)
